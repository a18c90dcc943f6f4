cd $GRAFT_REPO_ROOT
OUT=t1 TMO=900 bash tools/gpu_job.sh tests tests/test_l1_full_width_gpu.py tests/test_l1_cox_gpu.py tests/test_reference_gpu.py -k "L2560 or t2e or cox"
OUT=lo_l0 TMO=300 bash tools/gpu_job.sh bench --samples 500000 --loocv --snps 8000 --one-chrom --phenos 10 --l0-only --steps 1 --warmup 1 --no-cpu
OUT=lo_l1q TMO=300 bash tools/gpu_job.sh bench --samples 500000 --loocv --snps 51200 --bsize 100 --phenos 2 --steps 1 --warmup 0 --no-cpu
OUT=lo_l1b TMO=300 bash tools/gpu_job.sh bench --samples 500000 --loocv --snps 51200 --bsize 100 --phenos 1 --bt --prev 0.1 --steps 1 --warmup 0 --no-cpu
OUT=bt_or TMO=700 bash tools/gpu_job.sh bench --samples 500000 --snps 51200 --bsize 100 --phenos 4 --bt --prev 0.05,0.3,0.01,0.5 --steps 1 --warmup 0 --no-cpu --oracle-check --oracle-trait 2
nproc; free -g | head -2
