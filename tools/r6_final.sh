#!/bin/bash
# The evidence set of round 6 on its final build, one gpurun call (see profiles/README.md).  PMC traffic first, so that the bench lines that
# follow quote it (bench.py refuses a traffic file measured on other kernel sources): the default command, BASELINE configs[2]'s level-1 Gram,
# configs[3]'s level-1 unit with the block / trait counts of the sub-run (525 blocks, 4 traits).  Then the default bench line with its
# sub-records, kernel statistics, the kernel sequence of a batch, the leave-one-out level 0 and the device BGEN decoder under rocprofv3, the
# whole GPU test suite and the smoke test.  QUICK=1: traffic of the default command + bench only.
cd $GRAFT_REPO_ROOT
R=${ROUND:-r6}
P=profiles
OUT=fin_traffic bash tools/gpu_job.sh traffic 109 1 --steps 2 --warmup 1 --no-cpu --no-extra --no-disk | tail -2
cp gpurun_out/fin_traffic/traffic.json $P/${R}_traffic.json; cp gpurun_out/fin_traffic/pmc_fetch.md $P/${R}_pmc_fetch.md; cp gpurun_out/fin_traffic/pmc_write.md $P/${R}_pmc_write.md
if [ -z "$QUICK" ]; then
RX="k_l1_gram128|k_bed_prep_rows" OUT=fin_c3traffic bash tools/gpu_job.sh traffic 512 10 --samples 500000 --snps 500000 --phenos 10 --steps 1 --warmup 0 --no-cpu | tail -2
cp gpurun_out/fin_c3traffic/traffic.json $P/${R}_config3_traffic.json
RX="k_bt_|k_wgram|k_wsplit|k_wg_reduce|k_bed_prep_rows" OUT=fin_c4traffic bash tools/gpu_job.sh traffic 525 4 --samples 500000 --snps 51200 --bsize 100 --phenos 4 --bt --prev 0.05,0.3,0.01,0.5 --steps 1 --warmup 0 --no-cpu | tail -2
cp gpurun_out/fin_c4traffic/traffic.json $P/${R}_config4_traffic.json
fi
OUT=fin_bench TMO=1700 bash tools/gpu_job.sh bench
cp gpurun_out/fin_bench/bench_line.json $P/${R}_bench_line.json
[ -n "$QUICK" ] && exit 0
OUT=fin_stats bash tools/gpu_job.sh stats --steps 4 --warmup 1 --no-cpu --no-extra --no-disk | head -14
cp gpurun_out/fin_stats/kernel_stats.md $P/${R}_kernel_stats.md; cp gpurun_out/fin_stats/kernel_stats.csv $P/${R}_kernel_stats.csv
OUT=fin_seq bash tools/gpu_job.sh seq --steps 2 --warmup 1 --no-cpu --no-extra --no-disk | head -4
cp gpurun_out/fin_seq/batch_sequence.md $P/${R}_batch_sequence.md
OUT=fin_c3stats bash tools/gpu_job.sh stats --samples 500000 --snps 500000 --phenos 10 --steps 1 --warmup 1 --no-cpu --no-extra --no-disk | head -16
cp gpurun_out/fin_c3stats/kernel_stats.md $P/${R}_config3_kernel_stats.md
OUT=fin_loostats bash tools/gpu_job.sh stats --samples 500000 --loocv --snps 16000 --one-chrom --phenos 10 --l0-only --steps 1 --warmup 1 --no-cpu | head -12
cp gpurun_out/fin_loostats/kernel_stats.md $P/${R}_loocv_level0_500k_kernel_stats.md
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fin_bgen_raw -- python tools/bgen_dev_probe.py 500000 6144 3072 > gpurun_out/fin_bgen.log 2>&1
python tools/prof_summary.py gpurun_out/fin_bgen_raw gpurun_out/fin_bgen_stats.md > /dev/null; rm -rf gpurun_out/fin_bgen_raw
( grep "rep \|zlib\|wrote" gpurun_out/fin_bgen.log | cut -c1-220; echo; head -8 gpurun_out/fin_bgen_stats.md ) > $P/${R}_bgen_device_decoder.md; cat $P/${R}_bgen_device_decoder.md
# the decoder's walk when the phenotypes differ in their missing values (ten traits, 5 % each), and its instruction counters
BGEN_PROBE_MASK=10,0.05 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fin_bgenm_raw -- python tools/bgen_dev_probe.py 500000 3072 3072 > gpurun_out/fin_bgenm.log 2>&1
python tools/prof_summary.py gpurun_out/fin_bgenm_raw gpurun_out/fin_bgenm_stats.md > /dev/null; rm -rf gpurun_out/fin_bgenm_raw
( grep "per-trait\|rep 2" gpurun_out/fin_bgenm.log | cut -c1-220; echo; grep "kernel\|---\|k_bgen" gpurun_out/fin_bgenm_stats.md | head -6 ) > gpurun_out/fin_bgen_walk_masks.md; cat gpurun_out/fin_bgen_walk_masks.md
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --kernel-include-regex "k_bgen" --output-format csv -d gpurun_out/fin_bgenp_raw -- python tools/bgen_dev_probe.py 500000 3072 3072 > gpurun_out/fin_bgenp.log 2>&1
python tools/pmc_summary.py gpurun_out/fin_bgenp_raw gpurun_out/fin_bgen_pmc_insts.md > /dev/null; rm -rf gpurun_out/fin_bgenp_raw; head -12 gpurun_out/fin_bgen_pmc_insts.md
OUT=fin_tests TMO=1700 bash tools/gpu_job.sh tests
cp gpurun_out/fin_tests/pytest.log $P/${R}_pytest_gpu_final.log
OUT=fin_smoke bash tools/gpu_job.sh smoke
# round 6 additions: where the panel-128 Cholesky's workgroup time goes (s_memrealtime sums), binary-trait level 1 by kernel, what a large device
# allocation costs right after a release (the scrub the from-files figures must be read against), configs[2] from files with the driver's whole log
RG_C128_DBG=1 RG_PIPELINES=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu --no-extra --no-disk 2>&1 | grep "c128 phases" | tail -2 > $P/${R}_chol_phases.txt; cat $P/${R}_chol_phases.txt | cut -c1-300
OUT=fin_btstats bash tools/gpu_job.sh stats --samples 500000 --steps 1 --no-cpu --snps 51200 --bsize 100 --phenos 4 --bt --prev 0.05,0.3,0.01,0.5 --warmup 0 | head -12
cp gpurun_out/fin_btstats/kernel_stats.md $P/${R}_config4_level1_kernel_stats.md
( ./tools/bin/alloc_probe2 100; ./tools/bin/alloc_probe2 100 | head -3 ) > $P/${R}_alloc_probe.txt 2>&1; cat $P/${R}_alloc_probe.txt
python tools/r6_ingest.py "SLEEP=15,RG_TIMING=1" "" "SLEEP=15,RG_INGEST_MAP=0" "SLEEP=15,RG_INGEST_MAP=1" > gpurun_out/fin_ingest.log 2>&1
( cat gpurun_out/fin_ingest.log | cut -c1-600; echo; echo "--- the whole log of the first run:"; grep -v "^ \*\|Rsq =\|^$\|making predictions\|^phenotype " gpurun_out/r6_ingest/run0.log | cut -c1-200 ) > $P/${R}_e2e_config3.log; head -6 $P/${R}_e2e_config3.log | cut -c1-300
