#!/bin/bash
# round 3, call A: BASELINE configs[2] in full on one GPU (500,000 x 500,000 x 10 QT, resident) with the oracle check of phenotype 0;
# set-up primitives (allocation probe) and the from-files run of configs[1] as it stands at the start of the round
O=gpurun_out/r3a
mkdir -p $O
free -g | head -2 > $O/host.txt; nproc >> $O/host.txt; df -h /tmp | tail -1 >> $O/host.txt
( time timeout 900 python bench.py --samples 500000 --snps 500000 --phenos 10 --steps 1 --warmup 1 --no-cpu --oracle-check ) > $O/config3_full.log 2>&1
tail -1 $O/config3_full.log | cut -c1-3000
tools/bin/alloc_probe > $O/alloc1.txt 2>&1
PROBE_HOLD=1 tools/bin/alloc_probe 0.25 > $O/alloc2.txt 2>&1
tools/bin/alloc_probe 1 8 > $O/alloc3.txt 2>&1
cat $O/alloc1.txt $O/alloc2.txt $O/alloc3.txt
timeout 300 python tools/cli_e2e.py > $O/e2e.log 2>&1
cut -c1-700 $O/e2e.log
