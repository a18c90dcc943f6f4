#!/bin/bash
# Round 6, last call on the frozen sources: the three PMC traffic files (every one is keyed to csrc/rg_api.hip, which the round's last additions
# touched), the staged-ingest tests, then the default bench line that quotes the traffic.  Copy gpurun_out/fin_* into profiles/ afterwards.
cd $GRAFT_REPO_ROOT
R=r6; P=profiles
OUT=fin_traffic bash tools/gpu_job.sh traffic 109 1 --steps 2 --warmup 1 --no-cpu --no-extra --no-disk | tail -2
cp gpurun_out/fin_traffic/traffic.json $P/${R}_traffic.json; cp gpurun_out/fin_traffic/pmc_fetch.md $P/${R}_pmc_fetch.md; cp gpurun_out/fin_traffic/pmc_write.md $P/${R}_pmc_write.md
RX="k_l1_gram128|k_bed_prep_rows" OUT=fin_c3traffic bash tools/gpu_job.sh traffic 512 10 --samples 500000 --snps 500000 --phenos 10 --steps 1 --warmup 0 --no-cpu | tail -2
cp gpurun_out/fin_c3traffic/traffic.json $P/${R}_config3_traffic.json
RX="k_bt_|k_wgram|k_wsplit|k_wg_reduce|k_bed_prep_rows" OUT=fin_c4traffic bash tools/gpu_job.sh traffic 525 4 --samples 500000 --snps 51200 --bsize 100 --phenos 4 --bt --prev 0.05,0.3,0.01,0.5 --steps 1 --warmup 0 --no-cpu | tail -2
cp gpurun_out/fin_c4traffic/traffic.json $P/${R}_config4_traffic.json
timeout 600 python -m pytest tests/test_cli_gpu.py tests/test_step1_gpu.py -q -k "staged or step1" 2>&1 | tail -3
OUT=fin_bench TMO=1700 bash tools/gpu_job.sh bench
