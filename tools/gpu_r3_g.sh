#!/bin/bash
# round 3, call G: PMC counters of the prediction kernels (ring vs staged) on one level-0 batch at 500,000 samples x 10 phenotypes
O=$GRAFT_REPO_ROOT/gpurun_out/r3g
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
W="--samples 500000 --snps 32000 --phenos 10 --steps 1 --warmup 0 --no-cpu"
for v in ring staged; do
  if [ $v = staged ]; then export RG_PRED_STAGED=1; else unset RG_PRED_STAGED; fi
  RG_PIPELINES=1 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc1_$v -- python $GRAFT_REPO_ROOT/bench.py $W > $O/pmc1_$v.log 2>&1
  RG_PIPELINES=1 timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/pmc2_$v -- python $GRAFT_REPO_ROOT/bench.py $W > $O/pmc2_$v.log 2>&1
  RG_PIPELINES=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$v -- python $GRAFT_REPO_ROOT/bench.py $W > $O/st_$v.log 2>&1
  cd $GRAFT_REPO_ROOT
  python tools/pmc_summary.py /tmp/pmc1_$v $O/pmc_sq_$v.md > /dev/null 2>&1
  python tools/pmc_summary.py /tmp/pmc2_$v $O/pmc_lds_$v.md > /dev/null 2>&1
  python tools/prof_summary.py /tmp/st_$v $O/stats_$v.md > /dev/null 2>&1
  grep "k_l0_pred_i8" $O/pmc_sq_$v.md $O/pmc_lds_$v.md | cut -d'|' -f2-6
  grep "k_l0_pred\|k_l0_scale\|k_pk_tr\|k_l0_stats\|k_beta" $O/stats_$v.md
  cd /tmp
done
