#!/bin/bash
# final evidence of round 2: the complete -m gpu suite, the Step-1 profile set (bench line, kernel stats, PMC), the Step-2 line and its kernels
mkdir -p gpurun_out/r2fin
( time timeout 900 python -m pytest tests -m gpu -q ) > gpurun_out/r2fin/pytest.log 2>&1
tail -4 gpurun_out/r2fin/pytest.log
ROUND=r2 timeout 600 bash tools/collect_profiles.sh > gpurun_out/r2fin/collect.log 2>&1
tail -3 gpurun_out/r2fin/collect.log
timeout 200 python tools/step2_bench_line.py > gpurun_out/r2/r2_step2_bench_line.json 2> gpurun_out/r2fin/s2line.err
cut -c1-600 gpurun_out/r2/r2_step2_bench_line.json
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2fin/s2stats -- python $GRAFT_REPO_ROOT/tools/step2_bench_line.py > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r2fin/s2stats.err
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2fin/s2pmc -- python $GRAFT_REPO_ROOT/tools/step2_bench_line.py > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r2fin/s2pmc.err
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py gpurun_out/r2fin/s2stats gpurun_out/r2/r2_step2_kernel_stats.md > /dev/null
python tools/pmc_summary.py gpurun_out/r2fin/s2pmc gpurun_out/r2/r2_step2_pmc_mfma.md > /dev/null
head -8 gpurun_out/r2/r2_step2_kernel_stats.md
grep "k_xy_i8\|k_s2" gpurun_out/r2/r2_step2_pmc_mfma.md | head -12
rm -rf gpurun_out/r2fin/s2stats gpurun_out/r2fin/s2pmc
