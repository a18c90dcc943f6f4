#!/bin/bash
# round-2 GPU session A: full -m gpu suite (incl. the driver-vs-regenie tests), bench line, env sweeps, kernel stats
export TMPDIR=/tmp
mkdir -p gpurun_out/r2a
O=gpurun_out/r2a
( time timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -60 ) > $O/pytest.log 2>&1
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
for v in "RG_CHOL_FULLPAD=1" "RG_NBLK=55" "RG_NBLK=28" "RG_PIPELINES=1" "RG_NBLK=55 RG_CHOL_FULLPAD=1" "RG_PIPELINES=3"; do
  echo "== $v" >> $O/sweep.log
  env $v timeout 300 python bench.py --no-cpu --steps 10 2>>$O/sweep.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()}, d['roofline']['frac'])" >> $O/sweep.log
done
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_a -o r2a -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 3 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof_a -name "*kernel_stats*" -exec cp {} $O/ \;
ls -la /tmp/prof_a >> $O/sweep.log 2>&1
