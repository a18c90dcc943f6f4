#!/bin/bash
# round 3, call W: level-0 batch sizes against the block factorization's rounds (one wave per system, 1,024 per round)
O=gpurun_out/r3w
mkdir -p $O
for nb in 55 41 37 28; do
  ( RG_NBLK=$nb timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu --no-disk --no-extra ) > $O/bench_nblk_$nb.log 2>&1
  grep '^{' $O/bench_nblk_$nb.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('RG_NBLK=$nb', round(d['ms_per_step'],2), round(d['roofline']['frac'],4), {k:round(v['ms'],2) for k,v in d['kernels'].items() if v.get('ms')})"
done
