#!/usr/bin/env python
"""BASELINE configs[4] on its real input format: `regenie-amd --step 2 --qt --bgen` on a BGEN v1.2 file (layout 2, 8 bits per probability,
zlib) written here at the full sample count -- 500,000 samples x M variants, 10 quantitative phenotypes, 10 covariates -- from process
start to exit, with the shares of the run the driver's log gives (inflate + probabilities -> dosages on the host threads, PCIe, device),
next to regenie itself (oracle/_ref/regenie, when present) on a BOUNDED sample of the same data: a second BGEN file with the first
`ref_variants` variants, on which both programs run and their result lines are compared.

The LOCO predictions both programs read come from this driver's own step 1 on a small .bed of the same samples.
Usage (GPU box):  python tools/bgen_e2e.py [N=500000] [M=100000] [ref_variants=2000] [P=10] [C=10]
The .bgen is written by a pool of worker processes (one zlib stream per variant; 1.5 MB each at 500,000 samples)."""
import multiprocessing as mp
import os
import struct
import subprocess
import sys
import time
import zlib

import numpy as np

D = "/tmp/bgen_e2e"


def _variant_block(args):
    """One variant's identifying data + compressed genotype block (published BGEN v1.2 layout): hard calls from Binomial(2, maf) with 30 %
    of them smeared into genuine probabilities, 0.2 % missing."""
    j, n, chrom = args
    rng = np.random.default_rng(1000003 + j)
    maf = 0.02 + 0.48 * rng.random()
    g = (rng.random(n) < maf).astype(np.uint8) + (rng.random(n) < maf).astype(np.uint8)       # copies of the FIRST allele
    hom = np.where(g == 2, 255, 0).astype(np.int16)
    het = np.where(g == 1, 255, 0).astype(np.int16)
    sm = rng.random(n) < 0.3
    a = (rng.random(n) * 80).astype(np.int16)
    hom = np.where(sm & (g == 2), hom - a, hom)
    het = np.where(sm & (g == 2), het + a // 2, het)
    het = np.where(sm & (g == 1), het - a, het)
    hom = np.where(sm & (g == 1), hom + a // 2, hom)
    het = np.where(sm & (g == 0), het + a // 2, het)
    miss = rng.random(n) < 0.002
    probs = np.stack([np.where(miss, 0, hom), np.where(miss, 0, het)], axis=1).astype(np.uint8)
    blk = struct.pack("<IHBB", n, 2, 2, 2) + np.where(miss, 0x82, 0x02).astype(np.uint8).tobytes() + bytes([0, 8]) + probs.tobytes()
    z = zlib.compress(blk, 1)
    rec = bytearray()
    for s in ("", "v%d" % j, str(chrom)):
        b = s.encode()
        rec += struct.pack("<H", len(b)) + b
    rec += struct.pack("<IH", j + 1, 2)
    for al in ("A", "G"):
        rec += struct.pack("<I", 1) + al.encode()
    rec += struct.pack("<II", len(z) + 4, len(blk)) + z
    return bytes(rec)


def write_bgen(path, n, m, chroms, nproc):
    flags = 1 | (2 << 2)                                  # zlib, layout 2, no embedded sample identifiers
    header = struct.pack("<III", 20, m, n) + b"bgen" + struct.pack("<I", flags)
    t0 = time.time()
    with open(path, "wb") as fh, mp.Pool(nproc) as pool:
        fh.write(struct.pack("<I", len(header)) + header)
        for rec in pool.imap(_variant_block, ((j, n, chroms[j]) for j in range(m)), chunksize=8):
            fh.write(rec)
    return time.time() - t0


def main(N=500000, M=100000, MREF=2000, P=10, C=10):
    os.makedirs(D, exist_ok=True)
    nproc = max(1, min(224, (os.cpu_count() or 8) - 8))
    rng = np.random.default_rng(5)
    # sample files, phenotypes, covariates
    with open(D + "/x.sample", "w") as fh:
        fh.write("ID_1 ID_2 missing\n0 0 0\n" + "".join("%d %d 0\n" % (i + 1, i + 1) for i in range(N)))
    cov = rng.standard_normal((N, C))
    y = rng.standard_normal((N, P)) + 0.2 * cov[:, :1]
    with open(D + "/x.pheno", "w") as f2, open(D + "/x.covar", "w") as f3:
        f2.write("FID IID " + " ".join("Y%d" % (q + 1) for q in range(P)) + "\n")
        f3.write("FID IID " + " ".join("C%d" % (q + 1) for q in range(C)) + "\n")
        f2.write("".join("%d %d " % (i + 1, i + 1) + " ".join("%.6f" % v for v in y[i]) + "\n" for i in range(N)))
        f3.write("".join("%d %d " % (i + 1, i + 1) + " ".join("%.6f" % v for v in cov[i]) + "\n" for i in range(N)))
    # a small .bed of the same samples for step 1 (LOCO predictions for both programs)
    ms1 = 2200
    with open(D + "/s.bed", "wb") as fh:
        fh.write(b"\x6c\x1b\x01")
        fh.write(rng.integers(0, 256, size=(ms1, N // 4), dtype=np.uint8).tobytes())
    with open(D + "/s.bim", "w") as fh:
        fh.write("".join("%d\ts%d\t0\t%d\tA\tG\n" % (j // 100 + 1, j, j + 1) for j in range(ms1)))
    with open(D + "/s.fam", "w") as fh:
        fh.write("".join("%d %d 0 0 0 -9\n" % (i + 1, i + 1) for i in range(N)))
    chroms = [j * 22 // M + 1 for j in range(M)]
    tw = write_bgen(D + "/x.bgen", N, M, chroms, nproc)
    print("x.bgen: %d samples x %d variants, %.1f GB, written in %.0f s by %d processes" % (N, M, os.path.getsize(D + "/x.bgen") / 1e9, tw, nproc), flush=True)
    tw = write_bgen(D + "/r.bgen", N, MREF, [j * 22 // MREF + 1 for j in range(MREF)], nproc)
    print("r.bgen: the reference's bounded sample, %d variants, %.2f GB" % (MREF, os.path.getsize(D + "/r.bgen") / 1e9), flush=True)
    here = os.path.dirname(os.path.abspath(__file__))
    exe = os.path.join(here, "..", "regenie_amd", "bin", "regenie-amd")
    ref = os.path.join(here, "..", "oracle", "_ref", "regenie")
    t0 = time.time()
    r = subprocess.run([exe, "--step", "1", "--qt", "--bed", D + "/s", "--phenoFile", D + "/x.pheno", "--covarFile", D + "/x.covar", "--bsize", "100",
                        "--out", D + "/s1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    print("step 1 (LOCO predictions, %d SNPs): %.1f s" % (ms1, time.time() - t0), flush=True)
    common = ["--step", "2", "--qt", "--sample", D + "/x.sample", "--phenoFile", D + "/x.pheno", "--covarFile", D + "/x.covar", "--pred", D + "/s1_pred.list"]
    print("host: %s hardware threads, cpu.max %s" % (os.cpu_count(), open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a"), flush=True)
    variants = [("bsize 400", 400, {}), ("bsize 1000", 1000, {})]
    for extra in os.environ.get("BGEN_E2E_VARIANTS", "").split(";"):          # e.g. "RG_S2_PREP_THREADS=64;RG_BGEN_ZLIB=1"
        if extra:
            variants.append(("bsize 400 " + extra, 400, dict(kv.split("=") for kv in extra.split(","))))
    for name, bsz, env in variants:
        t0 = time.time()
        r = subprocess.run([exe] + common + ["--bgen", D + "/x.bgen", "--bsize", str(bsz), "--out", D + "/s2"], capture_output=True, text=True,
                           env=dict(os.environ, RG_TIMING="1", **env))
        dt = time.time() - t0
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        marks = [ln.strip() for ln in (r.stdout + r.stderr).split("\n") if "Elapsed" in ln or "since start" in ln or "[timing] step 2" in ln]
        import re
        chr_ms = [int(x) for x in re.findall(r"reading loco predictions for the chromosome\.\.\.done \((\d+)ms\)", r.stdout)]
        blk_ms = [int(x) for x in re.findall(r"block \[\d+/\d+\] : done \((\d+)ms\)", r.stdout)]
        if chr_ms and blk_ms:
            marks.append("chromosome set-ups %d x median %d ms (sum %d); blocks %d x median %d ms (sum %d)"
                         % (len(chr_ms), sorted(chr_ms)[len(chr_ms) // 2], sum(chr_ms), len(blk_ms), sorted(blk_ms)[len(blk_ms) // 2], sum(blk_ms)))
        print("regenie-amd --step 2 --bgen, %-28s: wall %.1f s = %.0f variants/s = %.2e variant*sample*pheno/s from the file | %s"
              % (name, dt, M / dt, M * N * P / dt, " | ".join(marks)), flush=True)
    # the bounded sample: both programs, line by line
    t0 = time.time()
    r = subprocess.run([exe] + common + ["--bgen", D + "/r.bgen", "--bsize", "400", "--out", D + "/r_amd"], capture_output=True, text=True)
    t_amd = time.time() - t0
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    if os.path.exists(ref):
        for thr in (16, 64):
            t0 = time.time()
            r = subprocess.run([ref] + common + ["--bgen", D + "/r.bgen", "--bsize", "400", "--threads", str(thr), "--out", D + "/r_ref"], capture_output=True, text=True)
            dt = time.time() - t0
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
            same = close = tot = 0
            for q in range(P):
                a = open(D + "/r_amd_Y%d.regenie" % (q + 1)).read().splitlines()
                b = open(D + "/r_ref_Y%d.regenie" % (q + 1)).read().splitlines()
                assert len(a) == len(b)
                tot += len(b) - 1
                same += sum(x == z for x, z in zip(a[1:], b[1:]))
                num = lambda ln: [t for k, t in enumerate(ln.split()) if k in (5, 6, 7, 9, 10, 11, 12)]     # A1FREQ INFO N | BETA SE CHISQ LOG10P
                close += sum(all(abs(float(u) - float(v)) <= 2e-5 * max(abs(float(v)), 1e-3) for u, v in zip(num(x), num(z)) if u != "NA" and v != "NA")
                             for x, z in zip(a[1:], b[1:]))
            print("bounded sample (%d variants x %d phenotypes): regenie v4.1.2 (oracle/_ref, --threads %d) %.1f s = %.0f variants/s; regenie-amd %.1f s; "
                  "%d of %d result lines byte-identical, %d within 2e-5" % (MREF, P, thr, dt, MREF / dt, t_amd, same, tot, close), flush=True)


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:]])
