cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=bi_tests TMO=600 bash tools/gpu_job.sh tests tests/test_bgen_device_gpu.py | tail -4
timeout 300 python tools/bgen_dev_probe.py 500000 3072 3072 2>&1 | grep "rep \|zlib" | cut -c1-200
