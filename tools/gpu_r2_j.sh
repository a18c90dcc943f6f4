#!/bin/bash
# round-2 GPU session J: pred_i8 with digit pairs; final default bench line
export TMPDIR=/tmp
O=gpurun_out/r2j
mkdir -p $O
( time timeout 900 python -m pytest tests/test_step1_gpu.py tests/test_reference_gpu.py tests/test_distributed_gpu.py -m gpu -q -x 2>&1 | tail -12 ) > $O/pytest.log 2>&1
timeout 600 python bench.py --samples 500000 --snps 62500 --phenos 10 --no-cpu --steps 3 --warmup 1 > $O/config3_share.json 2> $O/config3_share.err
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
