cd $GRAFT_REPO_ROOT
OUT=fin3_tests TMO=900 bash tools/gpu_job.sh tests tests/test_bgen_device_gpu.py tests/test_cli_gpu.py tests/test_step2_qt_gpu.py
OUT=fin3_bench TMO=1700 bash tools/gpu_job.sh bench
