cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ra
export BGEN_E2E_NCHR=2
( time timeout 900 python tools/bgen_e2e.py --json gpurun_out/ra/rec.json 500000 18432 1024 ) > gpurun_out/ra/e2e.log 2>&1
tail -40 gpurun_out/ra/e2e.log
