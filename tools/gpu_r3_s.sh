#!/bin/bash
# round 3, call S: Cox ridge level 1 at BASELINE configs[3]'s shape (500,000 samples, L = 2,560 level-0 predictors), one trait; multi-GPU t2e test
O=gpurun_out/r3s
mkdir -p $O
( time timeout 600 python -m pytest tests/test_cli_gpu.py -x -q -m gpu -k "t2e" ) > $O/pytest.log 2>&1
grep "passed\|failed\|error" $O/pytest.log | tail -3
( time timeout 1500 python bench.py --samples 500000 --snps 512000 --phenos 1 --t2e --steps 1 --warmup 0 --no-cpu --no-disk --no-extra ) > $O/t2e_500k.log 2>&1
grep '^{' $O/t2e_500k.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('level1'), {k:round(v['ms'],1) for k,v in d['kernels'].items() if v.get('ms')})" || tail -5 $O/t2e_500k.log
