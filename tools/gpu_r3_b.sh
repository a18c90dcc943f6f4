#!/bin/bash
# round 3, call B: the LDS-staged level-1 Gram (k_l1_gram128) -- parity tests, then BASELINE configs[2] in full with both kernels
O=gpurun_out/r3b
mkdir -p $O
( time timeout 900 python -m pytest tests/test_step1_gpu.py tests/test_l1_models_gpu.py tests/test_distributed_gpu.py tests/test_kernels_gpu.py -x -q -m gpu ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
( time timeout 600 python bench.py --samples 500000 --snps 500000 --phenos 10 --steps 1 --warmup 1 --no-cpu --oracle-check ) > $O/config3_new.log 2>&1
tail -4 $O/config3_new.log | cut -c1-2500
( time RG_L1_GRAM64=1 timeout 600 python bench.py --samples 500000 --snps 500000 --phenos 10 --steps 1 --warmup 1 --no-cpu ) > $O/config3_old.log 2>&1
tail -4 $O/config3_old.log | cut -c1-1800
timeout 300 python bench.py --no-cpu > $O/config2.log 2>&1
tail -1 $O/config2.log | cut -c1-1500
