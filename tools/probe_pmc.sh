#!/bin/bash
# LDS bank-conflict counters of the kernels that stage operands through LDS (one PMC pass, no API tracing):
#   SQ_LDS_BANK_CONFLICT = extra LDS cycles spent on conflicts, SQ_LDS_IDX_ACTIVE = all LDS-array cycles.
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/lds
rm -rf $O
RG_PIPELINES=1 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d $O -- python bench.py --steps 1 --warmup 0 --no-cpu > $O.log 2>&1
python tools/pmc_summary.py $O gpurun_out/${ROUND:-r1}_pmc_lds.md | grep -E "kernel|k_chol|k_gram_fp4|k_l0_pred|k_assemble|k_geno"
rm -rf $O
