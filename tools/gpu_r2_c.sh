#!/bin/bash
# round-2 GPU session C: driver (step 2, ingest policy), 500K oracle tests, bench line
export TMPDIR=/tmp
mkdir -p gpurun_out/r2c
O=gpurun_out/r2c
( time timeout 1500 python -m pytest tests/test_cli_gpu.py tests/test_reference_gpu.py tests/test_step2_qt_gpu.py -m gpu -q -s 2>&1 | tail -60 ) > $O/pytest.log 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
