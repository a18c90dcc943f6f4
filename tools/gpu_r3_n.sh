#!/bin/bash
# round 3, call N: the strip kernel with eight waves per workgroup (128 rows share one staging of the group's rows) against four; kernel tests,
# then the driver tests touched since call M (null-Firth files, gz LOCO in step 2, parallel phenotype parsing)
O=gpurun_out/r3n
mkdir -p $O
( time timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu ) > $O/pytest_kernels.log 2>&1
grep "passed\|failed\|error" $O/pytest_kernels.log | tail -3
for v in 0 1; do
  ( time RG_GSTRIP4=$v timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu --no-disk --no-extra ) > $O/bench_gstrip4_$v.log 2>&1
  grep '^{' $O/bench_gstrip4_$v.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('RG_GSTRIP4=$v', d['ms_per_step'], d['roofline']['frac'], {k:round(v['ms'],2) for k,v in d['kernels'].items() if v.get('ms')})"
done
cd /tmp && export TMPDIR=/tmp
( timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o gs8 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-disk --no-extra ) > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-160
find $O/prof -name '*.db' -delete; find $O/prof -name '*kernel_trace.csv' -size +20M -delete
( time timeout 1200 python -m pytest tests/test_step1_gpu.py tests/test_reference_gpu.py tests/test_cli_gpu.py -x -q -m gpu ) > $O/pytest_drivers.log 2>&1
grep "passed\|failed\|error" $O/pytest_drivers.log | tail -3
