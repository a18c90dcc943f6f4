#!/bin/bash
# BASELINE configs[1] from files: registered mapping (default) against the unregistered one
cd $GRAFT_REPO_ROOT
RG_E2E_WRITE_ONLY=1 python tools/cli_e2e.py 50000 100000 1 | tail -1
exe=regenie_amd/bin/regenie-amd
for v in "" "RG_INGEST_MAP=2" "" "RG_INGEST_MAP=2" "RG_INGEST_MAP=0" "RG_TIMING=1" "RG_INGEST_MAP=2 RG_TIMING=1"; do
  sleep 3
  s=$(date +%s.%N)
  env $v $exe --step 1 --bed /tmp/e2e/x --phenoFile /tmp/e2e/x.pheno --covarFile /tmp/e2e/x.covar --bsize 1000 --out /tmp/e2e/out > /tmp/e2e/run.log 2>&1
  e=$(date +%s.%N)
  echo "[$v] wall $(echo "$e - $s" | bc) | $(grep 'since start\|level 1 for\|timing\] rg_set_problem (GPU' /tmp/e2e/run.log | tr '\n' '|' | cut -c1-500)"
done
