cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=tp TMO=1200 bash tools/gpu_job.sh tests tests/test_step1_gpu.py tests/test_reference_gpu.py tests/test_cli_gpu.py -k "not 500 and not step2 and not bgen" 2>&1 | tail -5
OUT=c3 TMO=600 bash tools/gpu_job.sh bench --samples 500000 --snps 500000 --phenos 10 --steps 1 --warmup 1 --no-cpu 2>&1 | tail -14
