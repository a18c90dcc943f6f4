#!/bin/bash
# round 3, call O: what bounds the strip kernel -- the same launch with every K-phase MFMA issued twice (1) or every ring unit copied twice (2)
O=gpurun_out/r3o
mkdir -p $O
for v in 0 1 2; do
  ( time RG_GSTRIP_DBG=$v timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu --no-disk --no-extra ) > $O/bench_dbg_$v.log 2>&1
  grep '^{' $O/bench_dbg_$v.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('RG_GSTRIP_DBG=$v', d['ms_per_step'], d['roofline']['frac'], {k:round(v['ms'],2) for k,v in d['kernels'].items() if v.get('ms')})"
done
( time timeout 600 python -m pytest tests/test_reference_gpu.py::test_driver_write_and_use_null_firth -x -q -m gpu ) > $O/pytest.log 2>&1
grep "passed\|failed\|error" $O/pytest.log | tail -3
