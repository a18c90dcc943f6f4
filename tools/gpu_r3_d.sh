#!/bin/bash
# round 3, call D: ring-staged prediction kernel (k_l0_pred_i8_ring): parity tests, configs[2] in full (new vs register-staged)
O=gpurun_out/r3d
mkdir -p $O
( time timeout 900 python -m pytest tests/test_step1_gpu.py tests/test_reference_gpu.py -x -q -m gpu ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
( time timeout 600 python bench.py --samples 500000 --snps 500000 --phenos 10 --steps 1 --warmup 1 --no-cpu --oracle-check ) > $O/config3_new.log 2>&1
tail -4 $O/config3_new.log | cut -c1-2500
( time RG_PRED_STAGED=1 timeout 600 python bench.py --samples 500000 --snps 500000 --phenos 10 --steps 1 --warmup 1 --no-cpu ) > $O/config3_old.log 2>&1
tail -4 $O/config3_old.log | cut -c1-1800
