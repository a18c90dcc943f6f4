cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/cpmc -- python tools/chol_probe.py 1024 1375 1 > gpurun_out/cpmc.log 2>&1
PMC_BY_GRID=1 python tools/pmc_summary.py gpurun_out/cpmc gpurun_out/cpmc.md
