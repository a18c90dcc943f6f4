#!/bin/bash
# Round 6, first GPU session: the parity holes VERDICT r5 names, on the device.
#  (a) leave-one-out level 0 at block orders 1,024 / 1,000 / 960 x 20,000 samples against the oracle (tests/test_loocv_gpu.py) and
#      bench.py --loocv --oracle-check at 500,000 samples (level0_vs_oracle)
#  (b) the drawn cases with the DRIVER ON THE MI355X on every input format and the binary-trait corrections
#  (c) the binary-trait Step-2 end-to-end byte-identity figure on the device (tools/cli_e2e_step2.py ... bt=1)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6_parity
O=gpurun_out/r6_parity
( time timeout 900 python -m pytest tests/test_loocv_gpu.py -q -m gpu -x ) > $O/pytest_loocv.log 2>&1; tail -5 $O/pytest_loocv.log
( time timeout 900 python bench.py --samples 500000 --steps 1 --no-cpu --loocv --snps 8000 --one-chrom --phenos 10 --l0-only --warmup 1 --oracle-check ) > $O/loocv_l0.log 2>&1
grep '^{' $O/loocv_l0.log | tail -1 > $O/loocv_l0_line.json
python - <<PY
import json
try:
    d = json.load(open("$O/loocv_l0_line.json"))
    print("loocv l0: ms_per_step", d["ms_per_step"], "level0_vs_oracle", d["level1"].get("level0_vs_oracle"), "chol_f64", d["kernels"]["chol_f64"])
except Exception as e:
    print("no loocv line", e); print(open("$O/loocv_l0.log").read()[-2000:])
PY
export FUZZ_BUDGET_S=${FUZZ_BUDGET_S:-700}
( time FUZZ_DRIVER=1 FUZZ_BGEN=2 FUZZ_PGEN=1 FUZZ_PREP=2 FUZZ_BT_STEP2=2 timeout 1200 python tests/golden/fuzz_oracle_vs_reference.py 1 200 $O/fuzz_driver_all.md ) > $O/fuzz_all.log 2>&1; tail -4 $O/fuzz_all.log | cut -c1-400
( time FUZZ_BUDGET_S=300 FUZZ_DRIVER=1 FUZZ_ROUTES=bt_kfold FUZZ_BT_STEP2=2 timeout 600 python tests/golden/fuzz_oracle_vs_reference.py 1101 8 $O/fuzz_driver_bt5000.md ) > $O/fuzz_bt5000.log 2>&1; tail -3 $O/fuzz_bt5000.log | cut -c1-400
( time timeout 900 python tools/cli_e2e_step2.py 50000 100000 1000 1 ) > $O/step2_bt_e2e.log 2>&1; tail -12 $O/step2_bt_e2e.log | cut -c1-300
