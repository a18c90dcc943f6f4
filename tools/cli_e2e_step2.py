#!/usr/bin/env python
"""End-to-end wall time of `regenie-amd --step 2 --qt` on a BASELINE configs[1]-sized .bed ON DISK (50,000 samples x 100,000 SNPs, 22
chromosomes, 1 QT): step 1 of the same driver writes the LOCO files, then step 2 is timed from process start to exit -- file parsing,
.bed reads, PCIe, the score tests, the .regenie text.  Usage (GPU box): python tools/cli_e2e_step2.py [N=50000] [M=100000]"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch


def main(N=50000, M=100000, bs=1000, bt=0):
    d = "/tmp/e2e"
    os.makedirs(d, exist_ok=True)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(7)
    spc = [M // 22 + (1 if c < M % 22 else 0) for c in range(22)]
    t0 = time.time()
    ysum = torch.zeros(N, dtype=torch.float64, device=dev)
    with open(d + "/x.bed", "wb") as fh:
        fh.write(b"\x6c\x1b\x01")
        done = 0
        while done < M:
            n = min(4000, M - done)
            maf = 0.05 + 0.45 * torch.rand(n, 1, generator=g, device=dev)
            dd = (torch.rand(n, N, generator=g, device=dev) < maf).to(torch.uint8) + (torch.rand(n, N, generator=g, device=dev) < maf).to(torch.uint8)
            code = torch.where(dd == 2, torch.zeros_like(dd), torch.where(dd == 1, torch.full_like(dd, 2), torch.full_like(dd, 3)))
            c = code.view(n, N // 4, 4)
            packed = (c[:, :, 0] | (c[:, :, 1] << 2) | (c[:, :, 2] << 4) | (c[:, :, 3] << 6)).contiguous()
            fh.write(packed.cpu().numpy().tobytes())
            ysum += (dd[:3].double() - 2 * maf[:3]).sum(0)
            done += n
    with open(d + "/x.bim", "w") as fh:
        j = 0
        for c, k in enumerate(spc):
            for i in range(k):
                fh.write("%d\tv%d\t0\t%d\tA\tC\n" % (c + 1, j, i + 1))
                j += 1
    rng = np.random.default_rng(1)
    y = ysum.cpu().numpy()
    y = 0.4 * y / y.std() + rng.standard_normal(N)
    if bt:      # binary trait: liability threshold at prevalence 0.2
        y = (y > np.quantile(y, 0.8)).astype(np.float64)
    cov = rng.standard_normal((N, 2))
    with open(d + "/x.fam", "w") as f1, open(d + "/x.pheno", "w") as f2, open(d + "/x.covar", "w") as f3:
        f2.write("FID IID Y1\n")
        f3.write("FID IID C1 C2\n")
        for i in range(N):
            f1.write("%d %d 0 0 0 -9\n" % (i + 1, i + 1))
            f2.write(("%d %d %d\n" if bt else "%d %d %.8f\n") % (i + 1, i + 1, y[i]))
            f3.write("%d %d %.8f %.8f\n" % (i + 1, i + 1, cov[i, 0], cov[i, 1]))
    print("data set written in %.1f s (%.2f GB .bed)" % (time.time() - t0, os.path.getsize(d + "/x.bed") / 1e9), flush=True)
    del dd, code, c, packed
    torch.cuda.empty_cache()
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "regenie_amd", "bin", "regenie-amd")
    common = ["--bed", d + "/x", "--phenoFile", d + "/x.pheno", "--covarFile", d + "/x.covar"]
    t0 = time.time()
    trait = ["--bt"] if bt else ["--qt"]
    r = subprocess.run([exe, "--step", "1"] + trait + common + ["--bsize", str(bs), "--out", d + "/s1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    print("step 1: wall %.2f s" % (time.time() - t0), flush=True)
    runs = [("bsize 1000", 1000, {}), ("bsize 1000", 1000, {}), ("bsize 4000", 4000, {}), ("bsize 4000", 4000, {}), ("bsize 4000, fp64 route", 4000, {"RG_S2_DENSE": "1"})]
    extra = []
    if bt:
        runs = [("--firth --approx, bsize 1000", 1000, {}), ("--firth --approx, bsize 1000", 1000, {}), ("--firth --approx, bsize 4000", 4000, {})]
        extra = ["--firth", "--approx"]
    for name, bsz, env in runs:
        t0 = time.time()
        r = subprocess.run([exe, "--step", "2"] + trait + extra + common + ["--bsize", str(bsz), "--pred", d + "/s1_pred.list", "--out", d + "/s2"],
                           capture_output=True, text=True, env=dict(os.environ, **env))
        dt = time.time() - t0
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        el = [ln.strip() for ln in r.stdout.split("\n") if "Elapsed" in ln or "since start" in ln]
        nl = sum(1 for _ in open(d + "/s2_Y1.regenie")) - 1
        print("step 2 %-24s wall %.2f s = %.0f variants/s, %.2e genotype*pheno/s end to end (%d result lines) | %s"
              % (name, dt, M / dt, M * N / dt, nl, " | ".join(el)), flush=True)
    ref = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "_ref", "regenie")
    if os.path.exists(ref):   # regenie itself on the same files and the same LOCO predictions, on the box's host cores
        for thr in ((16,) if bt else (16, 64)):
            t0 = time.time()
            r = subprocess.run([ref, "--step", "2"] + trait + extra + common + ["--bsize", "1000", "--pred", d + "/s1_pred.list", "--threads", str(thr),
                               "--out", d + "/ref"], capture_output=True, text=True)
            dt = time.time() - t0
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
            a = open(d + "/s2_Y1.regenie").read().splitlines()
            b = open(d + "/ref_Y1.regenie").read().splitlines()
            same = sum(x == y for x, y in zip(a, b))
            close = sum(all(abs(float(u) - float(v)) <= 3e-4 * max(abs(float(v)), 1e-3) for u, v in zip(x.split()[8:12], y.split()[8:12]) if u != "NA" and v != "NA")
                        for x, y in zip(a[1:], b[1:]))
            print("regenie v4.1.2 (oracle/_ref, --threads %d): wall %.2f s = %.0f variants/s; %d of %d result lines byte-identical to regenie-amd's, %d of %d within 3e-4"
                  % (thr, dt, M / dt, same, len(b), close, len(b) - 1), flush=True)


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:]])
