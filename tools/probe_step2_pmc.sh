#!/bin/bash
# PMC passes over the Step-2 QT kernels (tools/step2_probe.py, one configuration): instruction mix / wait cycles, then HBM bytes.
# Counter passes carry --kernel-trace only (no API tracing), one counter group per pass.
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONPATH=$PWD
CFG=${S2_CFG:-200000,10,10,512}
for pass in "sq1:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "sq2:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "rd:FETCH_SIZE" "wr:WRITE_SIZE"; do
  tag=${pass%%:*}; ctr=${pass#*:}
  O=gpurun_out/s2pmc_$tag
  rm -rf $O
  timeout 60 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O -- python tools/step2_probe.py $CFG > $O.log 2>&1
  echo "== $tag ($ctr) rc=$?"
  python tools/pmc_summary.py $O gpurun_out/r1_step2_pmc_$tag.md | grep -E "kernel|k_s2"
  rm -rf $O
done
