#!/bin/bash
# round-2 GPU session I: complete -m gpu suite on the final kernels + PMC passes of the config-3 per-GPU share
export TMPDIR=/tmp
O=gpurun_out/r2i
mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $O/pytest.log 2>&1
W="--samples 500000 --snps 16000 --phenos 10 --no-cpu --steps 1 --warmup 0"
cd /tmp
RG_PIPELINES=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d /tmp/pmc1 -- python $GRAFT_REPO_ROOT/bench.py $W > /dev/null 2>&1
RG_PIPELINES=1 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/pmc2 -- python $GRAFT_REPO_ROOT/bench.py $W > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py /tmp/pmc1 $O/r2_config3_pmc_sq.md > /dev/null 2>&1
python tools/pmc_summary.py /tmp/pmc2 $O/r2_config3_pmc_lds.md > /dev/null 2>&1
