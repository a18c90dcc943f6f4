cd $GRAFT_REPO_ROOT
OUT=fin_traffic bash tools/gpu_job.sh traffic 109 1 --steps 2 --warmup 1 --no-cpu --no-extra --no-disk | tail -3
cp gpurun_out/fin_traffic/traffic.json profiles/r4_traffic.json
cp gpurun_out/fin_traffic/pmc_fetch.md profiles/r4_pmc_fetch.md; cp gpurun_out/fin_traffic/pmc_write.md profiles/r4_pmc_write.md
OUT=fin_bench TMO=1700 bash tools/gpu_job.sh bench
OUT=fin_stats bash tools/gpu_job.sh stats --steps 4 --warmup 1 --no-cpu --no-extra --no-disk | head -16
OUT=fin_seq bash tools/gpu_job.sh seq --steps 2 --warmup 1 --no-cpu --no-extra --no-disk | head -5
OUT=fin_c3stats bash tools/gpu_job.sh stats --samples 500000 --snps 500000 --phenos 10 --steps 1 --warmup 1 --no-cpu --no-extra --no-disk | head -22
OUT=fin_tests TMO=1500 bash tools/gpu_job.sh tests
OUT=fin_smoke bash tools/gpu_job.sh smoke
cp profiles/r4_traffic.json gpurun_out/fin_traffic/r4_traffic.json
