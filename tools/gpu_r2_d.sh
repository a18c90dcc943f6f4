#!/bin/bash
# round-2 GPU session D: complete -m gpu suite, driver end-to-end variants, batch / pipeline sweep, config-3 per-GPU share
export TMPDIR=/tmp
mkdir -p gpurun_out/r2d
O=gpurun_out/r2d
( time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $O/pytest.log 2>&1
timeout 400 python tools/cli_e2e.py > $O/e2e.log 2>&1
for v in "RG_NBLK=64" "RG_NBLK=40 RG_PIPELINES=3" "RG_NBLK=28 RG_PIPELINES=4" "RG_NBLK=37 RG_PIPELINES=3"; do
  echo "== $v" >> $O/sweep.log
  env $v timeout 300 python bench.py --no-cpu --steps 10 2>>$O/sweep.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()}, d['roofline']['frac'])" >> $O/sweep.log
done
timeout 600 python bench.py --samples 500000 --snps 62500 --phenos 10 --no-cpu --steps 3 --warmup 1 > $O/config3_share.json 2> $O/config3_share.err
