#!/bin/bash
# round-2 GPU session B: driver changes (streamed ingest, multi-GPU host), bench with the from-files leg, profiles
export TMPDIR=/tmp
mkdir -p gpurun_out/r2b
O=gpurun_out/r2b
( time timeout 1200 python -m pytest tests/test_cli_gpu.py tests/test_reference_gpu.py tests/test_distributed_gpu.py tests/test_step1_gpu.py tests/test_loocv_gpu.py tests/test_l0_f64_gpu.py tests/test_fullsize_gpu.py tests/test_abi.py -m gpu -x -q -s 2>&1 | tail -40 ) > $O/pytest.log 2>&1
timeout 120 python tools/ref_threads_probe.py 8 16 32 64 > $O/ref_threads.log 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
ROUND=r2 timeout 900 bash tools/collect_profiles.sh > $O/collect.log 2>&1
