#!/bin/bash
# BASELINE configs[1] from files, runs back to back and runs with a pause: does a run wait for what the previous one released?
cd $GRAFT_REPO_ROOT
RG_E2E_WRITE_ONLY=1 python tools/cli_e2e.py 50000 100000 1 | tail -1
exe=regenie_amd/bin/regenie-amd
for pause in 0 0 0 5 5 0 2 2; do
  sleep $pause
  s=$(date +%s.%N)
  RG_TIMING=1 $exe --step 1 --bed /tmp/e2e/x --phenoFile /tmp/e2e/x.pheno --covarFile /tmp/e2e/x.covar --bsize 1000 --out /tmp/e2e/out > /tmp/e2e/run.log 2>&1
  e=$(date +%s.%N)
  echo "[pause $pause] wall $(echo "$e - $s" | bc) | $(grep 'since start\|timing\] rg_set_problem' /tmp/e2e/run.log | tr '\n' '|' | cut -c1-700)"
done
