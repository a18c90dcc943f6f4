cd $GRAFT_REPO_ROOT
OUT=fin_traffic bash tools/gpu_job.sh traffic 109 1 --steps 2 --warmup 1 --no-cpu --no-extra --no-disk | tail -2
cp gpurun_out/fin_traffic/traffic.json profiles/r5_traffic.json
RX="k_l1_gram128|k_bed_prep_rows" OUT=fin_c3traffic bash tools/gpu_job.sh traffic 512 10 --samples 500000 --snps 500000 --phenos 10 --steps 1 --warmup 0 --no-cpu | tail -1
cp gpurun_out/fin_c3traffic/traffic.json profiles/r5_config3_traffic.json
RX="k_bt_|k_wgram|k_wsplit|k_wg_reduce|k_bed_prep_rows" OUT=fin_c4traffic bash tools/gpu_job.sh traffic 525 4 --samples 500000 --snps 51200 --bsize 100 --phenos 4 --bt --prev 0.05,0.3,0.01,0.5 --steps 1 --warmup 0 --no-cpu | tail -1
cp gpurun_out/fin_c4traffic/traffic.json profiles/r5_config4_traffic.json
OUT=fin_bench TMO=1700 bash tools/gpu_job.sh bench
OUT=fin_tests TMO=1700 bash tools/gpu_job.sh tests
OUT=fin_smoke bash tools/gpu_job.sh smoke
