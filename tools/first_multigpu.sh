#!/bin/bash
# First contact with a multi-GPU node (no such box was available to rounds 1 - 3: RCCL has only ever run with a world of one, the
# multi-rank logic over peer copies on one device).  Run from the repository root on a node with N >= 2 MI355X:
#   bash tools/first_multigpu.sh [N=8]
# 1. the C++ driver's multi-GPU tests WITHOUT --single-device: real devices, RCCL transport, .loco files byte-identical to one GPU
# 2. an N-GPU == 1-GPU byte-identity check of the driver on a 50,000 x 100,000 data set written to /tmp, RCCL and peer transports
# 3. bench.py --gpus 2/4/N: BASELINE configs[2] under strong scaling (one process per GPU, torch.distributed over RCCL)
set -u
N=${1:-8}
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/multigpu
mkdir -p $O
python -c "import __graft_entry__ as g; g.build()"
echo "== 1. driver tests on real devices (RG_TEST_REAL_GPUS=1 drops --single-device / --transport peer)"
RG_TEST_REAL_GPUS=1 python -m pytest tests/test_cli_gpu.py -q -m gpu -k "multi_gpu" 2>&1 | tail -5 | tee $O/pytest_multi.log
echo "== 2. N-GPU == 1-GPU byte identity from files"
python - <<PY 2>&1 | tee $O/identity.log
import filecmp, os, subprocess, sys
sys.path.insert(0, ".")
sys.argv = ["cli_e2e", "50000", "100000"]
exe = os.path.abspath("regenie_amd/bin/regenie-amd")
d = "/tmp/e2e"
if not os.path.exists(d + "/x.bed"):
    import tools.cli_e2e as e
    try:
        e.main(50000, 100000)
    except SystemExit:
        pass
base = [exe, "--step", "1", "--bed", d + "/x", "--phenoFile", d + "/x.pheno", "--covarFile", d + "/x.covar", "--bsize", "1000"]
subprocess.run(base + ["--out", d + "/one"], check=True, capture_output=True)
for n in (2, 4, $N):
    for tr in ("rccl", "peer"):
        for extra in ([], ["--l1-shared"]):
            r = subprocess.run(base + ["--gpus", str(n), "--transport", tr] + extra + ["--out", d + "/multi"], capture_output=True, text=True)
            same = r.returncode == 0 and filecmp.cmp(d + "/one_1.loco", d + "/multi_1.loco", shallow=False)
            wall = [ln for ln in r.stdout.splitlines() if "Elapsed" in ln]
            print("gpus %d %-4s %-12s rc %d identical %s %s" % (n, tr, " ".join(extra), r.returncode, same, wall[-1:] ))
PY
echo "== 3. bench.py strong scaling of BASELINE configs[2]"
for n in 1 2 4 $N; do
  if [ $n = 1 ]; then python bench.py --samples 500000 --snps 500000 --phenos 10 --steps 2 --warmup 1 --no-cpu > $O/bench_$n.json 2> $O/bench_$n.err
  else python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus $n --steps 2 --warmup 1 > $O/bench_$n.json 2> $O/bench_$n.err; fi
  tail -1 $O/bench_$n.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['n_gpus'], 'GPUs', round(d['ms_per_step'],1), 'ms/step', '%.3e' % d['value'], d['config']['parallelism'])"
done
echo "== 4. round 5's entry points on several devices: --step 1 --loocv (the tridiagonal level 0) and --step 2 --bgen (one device decoder per part)"
python - <<PY 2>&1 | tee $O/round5.log
import filecmp, os, subprocess
exe = os.path.abspath("regenie_amd/bin/regenie-amd")
d = "/tmp/e2e"
base = [exe, "--step", "1", "--bed", d + "/x", "--phenoFile", d + "/x.pheno", "--covarFile", d + "/x.covar", "--bsize", "1000", "--loocv"]
subprocess.run(base + ["--out", d + "/loo1"], check=True, capture_output=True)
for n in (2, $N):
    r = subprocess.run(base + ["--gpus", str(n), "--out", d + "/loon"], capture_output=True, text=True)
    print("loocv gpus %d rc %d identical %s" % (n, r.returncode, r.returncode == 0 and filecmp.cmp(d + "/loo1_1.loco", d + "/loon_1.loco", shallow=False)))
PY
# the BGEN Step 2 of tools/bgen_e2e.py on N devices (contiguous block ranges per GPU, one decoder each, part files concatenated in block order); its bounded sample
# is compared line by line with regenie itself
BGEN_E2E_NCHR=4 BGEN_E2E_ARGS="--gpus $N" python tools/bgen_e2e.py 500000 36864 1024 2>&1 | grep "regenie-amd --step 2\|bounded sample" | cut -c1-300 | tee -a $O/round5.log
