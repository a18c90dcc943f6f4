#!/usr/bin/env python
"""Times oracle/_ref/regenie (the reference itself) --step 1 on a 50,000 x 2,000 synthetic .bed at several --threads values:
which thread count the cpu_baseline leg of bench.py should give the reference on this box (its own default is all cores - 1,
Regenie.cpp:1104-1106; Eigen's GEMM does not scale to hundreds of threads)."""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = os.path.join(ROOT, "oracle", "_ref", "regenie")
N, M = 50000, 2000
d = tempfile.mkdtemp(prefix="rg_thr_")
rng = np.random.default_rng(1)
maf = 0.05 + 0.45 * rng.random(M)
with open(d + "/s.bed", "wb") as fh:
    fh.write(b"\x6c\x1b\x01")
    for j in range(M):
        g = (rng.random(N) < maf[j]).astype(np.int8) + (rng.random(N) < maf[j]).astype(np.int8)
        c = np.where(g == 2, 0, np.where(g == 1, 2, 3)).astype(np.uint8).reshape(-1, 4)
        fh.write((c[:, 0] | (c[:, 1] << 2) | (c[:, 2] << 4) | (c[:, 3] << 6)).astype(np.uint8).tobytes())
open(d + "/s.bim", "w").write("".join("%d\ts%d\t0\t%d\tA\tG\n" % (1 + j // 1000, j, j + 1) for j in range(M)))
open(d + "/s.fam", "w").write("".join("%d %d 0 0 0 -9\n" % (i + 1, i + 1) for i in range(N)))
y = rng.standard_normal(N)
open(d + "/s.pheno", "w").write("FID IID Y1\n" + "".join("%d %d %.10g\n" % (i + 1, i + 1, y[i]) for i in range(N)))
for thr in [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64, 128]:
    t0 = time.time()
    r = subprocess.run([REG, "--step", "1", "--bed", d + "/s", "--phenoFile", d + "/s.pheno", "--bsize", "1000", "--qt", "--threads", str(thr),
                        "--out", d + "/o"], capture_output=True, text=True)
    ph = [ln.strip() for ln in r.stdout.splitlines() if "done (" in ln and ("working" in ln or "level 0 ridge" in ln)]
    print("threads", thr, "wall %.1f s" % (time.time() - t0), ph[:2], flush=True)
