#!/usr/bin/env python
"""End-to-end wall time of the C++ driver on a synthetic BGEN v1.2 file (8-bit, zlib): inflate + dosage decode on the host
threads, fp64 level 0 on the GPU.  Usage (GPU box): python tests/e2e_bgen.py [N=20000] [M=4000]"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

from oracle import bgen as obg   # fixture writer only: this measurement script lives under tests/ for that reason


def main(N=20000, M=4000, bs=1000):
    d = "/tmp/bge2e"
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(3)
    t0 = time.time()
    maf = rng.uniform(0.05, 0.5, M)
    hard = (rng.random((M, N)) < maf[:, None]).astype(np.int16) + (rng.random((M, N)) < maf[:, None]).astype(np.int16)
    p0 = np.clip(np.where(hard == 2, 235, np.where(hard == 1, 15, 5)) + rng.integers(-5, 6, (M, N)), 0, 255)
    p1 = np.clip(np.where(hard == 1, 230, 12) + rng.integers(-5, 6, (M, N)), 0, 255 - p0)
    probs = np.stack([p0, p1], axis=-1).astype(np.uint8)
    miss = rng.random((M, N)) < 0.002
    variants = [(1 + (j * 22) // M, j + 1, "v%d" % j, "A", "C") for j in range(M)]
    obg.write_bgen(d + "/x.bgen", probs, miss, variants, sample_ids=["%d_%d" % (i + 1, i + 1) for i in range(N)], compression=1)
    dos = (p1 + 2.0 * p0) / 255.0
    y = 0.5 * (dos[:5] - dos[:5].mean(1, keepdims=True)).sum(0) + rng.standard_normal(N)
    with open(d + "/x.pheno", "w") as f2:
        f2.write("FID IID Y1\n")
        for i in range(N):
            f2.write("%d %d %.8f\n" % (i + 1, i + 1, y[i]))
    print("bgen written in %.1f s (%.1f MB)" % (time.time() - t0, os.path.getsize(d + "/x.bgen") / 1e6), flush=True)
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "regenie_amd", "bin", "regenie-amd")
    for rep in range(2):
        t0 = time.time()
        r = subprocess.run([exe, "--step", "1", "--bgen", d + "/x.bgen", "--phenoFile", d + "/x.pheno", "--bsize", str(bs), "--out", d + "/out"],
                           capture_output=True, text=True)
        dt = time.time() - t0
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        l0 = [ln for ln in r.stdout.split("\n") if "level 0 ridge on GPU" in ln]
        gpu_ms = sum(int(ln.split("level 0 ridge on GPU ")[1].split("ms")[0]) for ln in l0)
        rd_ms = sum(int(ln.split("(read ")[1].split("ms")[0]) for ln in l0)
        print("run %d: wall %.2f s = %.2e SNP*sample/s end to end; inflate + decode %d ms, level 0 (PCIe + GPU, fp64 path) %d ms" % (
            rep, dt, M * N / dt, rd_ms, gpu_ms), flush=True)


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:]])
