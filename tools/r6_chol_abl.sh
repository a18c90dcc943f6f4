#!/bin/bash
# ablations of the panel-128 Cholesky (timing only; flags give wrong results): where does the slot time go
cd $GRAFT_REPO_ROOT
for f in ${FLAGS:-0}; do
  echo "== RG_C128_FLAGS=$f"
  RG_C128_FLAGS=$f RG_C128_DBG=1 RG_PIPELINES=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu --no-extra --no-disk --l0-only 2>&1 | grep "c128 phases\|Error\|error" | tail -1 | cut -c1-1200
  RG_C128_FLAGS=$f timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu --no-extra --no-disk --l0-only 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'chol ms', d['kernels']['chol_f64']['ms'], d['roofline']['frac'])"
done
