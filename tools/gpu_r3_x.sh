#!/bin/bash
# round 3, call X: k_chol_gfact held to 256 registers (two waves per SIMD allowed: waves of the other pipeline's kernels can share its SIMDs)
O=gpurun_out/r3x
mkdir -p $O
( timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k chol ) > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for it in 1 2; do
( timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu --no-disk --no-extra ) > $O/bench_$it.log 2>&1
grep '^{' $O/bench_$it.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), round(d['roofline']['frac'],4), {k:round(v['ms'],2) for k,v in d['kernels'].items() if v.get('ms')})"
done
