#!/bin/bash
# round 3, call E: (1) per-column factorization of the group blocks (k_chol_diag / k_chol_panel in place of k_chol_gfact): parity, config-2 bench
# both ways; (2) kernel stats of a config-3 share (64,000 SNPs x 500,000 samples x 10 phenotypes), single pipeline, ring vs staged predictions
O=gpurun_out/r3e
mkdir -p $O
( time timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_step1_gpu.py tests/test_loocv_gpu.py tests/test_l1_models_gpu.py -x -q -m gpu ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 300 python bench.py --no-cpu > $O/config2_new.log 2>&1
tail -1 $O/config2_new.log | cut -c1-200; tail -1 $O/config2_new.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:round(v['ms'],2) for k,v in d['kernels'].items()}, d['roofline']['frac'], d['loco_checksum'])"
RG_CHOL_GFACT=1 timeout 300 python bench.py --no-cpu > $O/config2_gfact.log 2>&1
tail -1 $O/config2_gfact.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:round(v['ms'],2) for k,v in d['kernels'].items()}, d['roofline']['frac'], d['loco_checksum'])"
cd /tmp && export TMPDIR=/tmp
for v in ring staged; do
  if [ $v = staged ]; then export RG_PRED_STAGED=1; else unset RG_PRED_STAGED; fi
  RG_PIPELINES=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_$v -- python $GRAFT_REPO_ROOT/bench.py --samples 500000 --snps 64000 --phenos 10 --steps 2 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/$O/stats_$v.log 2>&1
  python $GRAFT_REPO_ROOT/tools/prof_summary.py $GRAFT_REPO_ROOT/$O/stats_$v $GRAFT_REPO_ROOT/$O/share_kernel_stats_$v.md > /dev/null
  head -24 $GRAFT_REPO_ROOT/$O/share_kernel_stats_$v.md | cut -c1-120
  rm -rf $GRAFT_REPO_ROOT/$O/stats_$v
done
