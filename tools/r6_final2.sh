#!/bin/bash
# Round 6, second half of the evidence set on the final build (the first half is `QUICK=1 bash tools/r6_final.sh`: PMC traffic of the default
# command + the default bench line): kernel statistics and the kernel sequence of a batch, configs[2]'s kernel statistics, the whole GPU test
# suite, the smoke test, configs[2] from files with the staged .bed, the masked hard-call block of Step 2 kernel by kernel, the Cholesky's
# phase / slot diagnostic.  Everything lands under gpurun_out/fin_* (copied into profiles/ by hand: the box's own profiles/ does not come back).
cd $GRAFT_REPO_ROOT
OUT=fin_stats bash tools/gpu_job.sh stats --steps 4 --warmup 1 --no-cpu --no-extra --no-disk | head -14
OUT=fin_seq bash tools/gpu_job.sh seq --steps 2 --warmup 1 --no-cpu --no-extra --no-disk | head -4
OUT=fin_c3stats bash tools/gpu_job.sh stats --samples 500000 --snps 500000 --phenos 10 --steps 1 --warmup 1 --no-cpu --no-extra --no-disk | head -16
OUT=fin_tests TMO=1700 bash tools/gpu_job.sh tests
OUT=fin_smoke bash tools/gpu_job.sh smoke
RG_TIMING=1 timeout 600 python tools/r6_ingest.py "" "SLEEP=12" "SLEEP=12" "SLEEP=12,RG_INGEST_STAGE=0" > gpurun_out/fin_ingest.txt 2>&1
( for f in gpurun_out/r6_ingest/run[123].log; do echo "== $f"; grep -v "^\[timing\] .loco\|^\[timing\] level 1 of" $f; done ) > gpurun_out/fin_e2e_config3_staged.log; rm -rf /tmp/e2e
grep "wall\|device memory\|predictions written" gpurun_out/fin_e2e_config3_staged.log | cut -c1-200
bash tools/r6_s2seq.sh > gpurun_out/fin_step2_masked_sequence.md 2>&1; cat gpurun_out/fin_step2_masked_sequence.md
FLAGS=0 bash tools/r6_chol_abl.sh > gpurun_out/fin_chol_phases.txt 2>&1; tr "|" "\n" < gpurun_out/fin_chol_phases.txt | tail -8
