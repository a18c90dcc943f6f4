#!/bin/bash
# The evidence set of the round on its final build, one gpurun call (see profiles/README.md): PMC traffic of the default command first, so that
# the bench line that follows quotes it (bench.py refuses a traffic file measured on other kernel sources), then kernel statistics, the
# kernel sequence of a batch, configs[2]'s statistics, the whole GPU test suite and the smoke test.  QUICK=1: traffic + bench + targeted tests only.
cd $GRAFT_REPO_ROOT
OUT=fin_traffic bash tools/gpu_job.sh traffic 109 1 --steps 2 --warmup 1 --no-cpu --no-extra --no-disk | tail -3
cp gpurun_out/fin_traffic/traffic.json profiles/r4_traffic.json
cp profiles/r4_traffic.json gpurun_out/fin_traffic/r4_traffic.json
OUT=fin_bench TMO=1700 bash tools/gpu_job.sh bench
if [ -n "$QUICK" ]; then
  OUT=fin_tests_quick bash tools/gpu_job.sh tests tests/test_l1_full_width_gpu.py tests/test_cli_gpu.py tests/test_l1_models_gpu.py -k "bt or quasi or L2560"
  exit 0
fi
OUT=fin_stats bash tools/gpu_job.sh stats --steps 4 --warmup 1 --no-cpu --no-extra --no-disk | head -16
OUT=fin_seq bash tools/gpu_job.sh seq --steps 2 --warmup 1 --no-cpu --no-extra --no-disk | head -5
OUT=fin_c3stats bash tools/gpu_job.sh stats --samples 500000 --snps 500000 --phenos 10 --steps 1 --warmup 1 --no-cpu --no-extra --no-disk | head -22
OUT=fin_tests TMO=1500 bash tools/gpu_job.sh tests
OUT=fin_smoke bash tools/gpu_job.sh smoke
