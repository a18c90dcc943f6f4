cd $GRAFT_REPO_ROOT
OUT=t5 TMO=900 bash tools/gpu_job.sh tests tests/test_kernels_gpu.py tests/test_loocv_gpu.py tests/test_l0_f64_gpu.py tests/test_reference_gpu.py -k "dgemm or loocv or loo or f64 or refcmd"
OUT=lo_l0e TMO=300 bash tools/gpu_job.sh bench --samples 500000 --loocv --snps 16000 --one-chrom --phenos 10 --l0-only --steps 1 --warmup 1 --no-cpu
OUT=lo_l0e_stats TMO=300 bash tools/gpu_job.sh stats --samples 500000 --loocv --snps 16000 --one-chrom --phenos 10 --l0-only --steps 1 --warmup 1 --no-cpu
