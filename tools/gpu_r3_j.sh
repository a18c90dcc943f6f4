#!/bin/bash
# round 3, call J: LDS-staged weighted Gram (k_wgram128): level-1 model tests, then BASELINE configs[3]'s pieces on one GPU --
# level 1 of 2 binary traits at the full L = 2,560 / 500,000 samples (both kernels), level 0 of a 64-block share at 50 phenotypes
O=gpurun_out/r3j
mkdir -p $O
( time timeout 900 python -m pytest tests/test_l1_models_gpu.py -x -q -m gpu ) > $O/pytest.log 2>&1
grep "passed\|failed\|error" $O/pytest.log | tail -3
( time timeout 900 python bench.py --samples 500000 --snps 500000 --phenos 2 --bt --steps 1 --warmup 0 --no-cpu ) > $O/config4_l1_new.log 2>&1
grep '^{' $O/config4_l1_new.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new', d['ms_per_step'], d['level1'], d['selected_tau_index'], d['loco_checksum'])"
( time RG_WGRAM64=1 timeout 900 python bench.py --samples 500000 --snps 500000 --phenos 2 --bt --steps 1 --warmup 0 --no-cpu ) > $O/config4_l1_old.log 2>&1
grep '^{' $O/config4_l1_old.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('old', d['ms_per_step'], d['level1'], d['selected_tau_index'], d['loco_checksum'])"
( time timeout 600 python bench.py --samples 500000 --snps 62500 --phenos 50 --bt --l0-only --steps 2 --warmup 1 --no-cpu ) > $O/config4_l0_share.log 2>&1
grep '^{' $O/config4_l0_share.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('l0 share', d['ms_per_step'], {k:round(v['ms'],1) for k,v in d['kernels'].items()})"
