#!/bin/bash
# round 3, call P: right-hand sides embedded in the systems' padding rows (no RHS tile row) against a tile row of their own
O=gpurun_out/r3p
mkdir -p $O
( time timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_step1_gpu.py tests/test_cli_gpu.py -x -q -m gpu ) > $O/pytest.log 2>&1
grep "passed\|failed\|error" $O/pytest.log | tail -3
for v in 0 1; do
  ( time RG_NO_EMBED=$v timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu --no-disk --no-extra --oracle-check ) > $O/bench_noembed_$v.log 2>&1
  grep '^{' $O/bench_noembed_$v.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('RG_NO_EMBED=$v', d['ms_per_step'], d['roofline']['frac'], {k:round(v['ms'],2) for k,v in d['kernels'].items() if v.get('ms')}, d['cpu_baseline'].get('full_config_vs_oracle'))"
done
