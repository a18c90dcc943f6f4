#!/usr/bin/env python
"""End-to-end wall time of the C++ driver on a BASELINE configs[1]-sized data set ON DISK (50,000 samples x 100,000 SNPs, 22
chromosomes, 1 QT, bsize 1000): file parsing, .bed reads, PCIe, level 0, level 1, .loco writing -- everything `bench.py`
leaves out on purpose.  Usage (GPU box): python tools/cli_e2e.py [N=50000] [M=100000] [P=1]
(N = M = 500000, P = 10 is BASELINE configs[2]: a 62.5 GB .bed; the variants then are the default and more reader threads only)"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch


def main(N=50000, M=100000, P=1, bs=1000):
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
    y = 0.4 * (y / y.std())[:, None] + rng.standard_normal((N, P))
    cov = rng.standard_normal((N, 2))
    with open(d + "/x.fam", "w") as f1, open(d + "/x.pheno", "w") as f2, open(d + "/x.covar", "w") as f3:
        f2.write("FID IID " + " ".join("Y%d" % (q + 1) for q in range(P)) + "\n")
        f3.write("FID IID C1 C2\n")
        for i in range(N):
            f1.write("%d %d 0 0 0 -9\n" % (i + 1, i + 1))
            f2.write("%d %d " % (i + 1, i + 1) + " ".join("%.8f" % v for v in y[i]) + "\n")
            f3.write("%d %d %.8f %.8f\n" % (i + 1, i + 1, cov[i, 0], cov[i, 1]))
    print("data set written in %.1f s (%.2f GB .bed)" % (time.time() - t0, os.path.getsize(d + "/x.bed") / 1e9), flush=True)
    del dd, code, c, packed
    torch.cuda.empty_cache()
    if os.environ.get("RG_E2E_WRITE_ONLY"):      # (tools/r6_ingest.py: the data set alone)
        return
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "regenie_amd", "bin", "regenie-amd")
    variants = [("default", {}), ("default", {}), ("default", {}), ("RG_NBLK=28", {"RG_NBLK": "28"}), ("RG_NBLK=28", {"RG_NBLK": "28"}),
                ("RG_PIPELINES=1", {"RG_PIPELINES": "1"}), ("RG_PIPELINES=1", {"RG_PIPELINES": "1"}),
                ("RG_INGEST_PINNED=1", {"RG_INGEST_PINNED": "1"}), ("sleep 3 s first", {})]
    if N * M > 2e10:      # a large input: the wall is the file read -- the default and more reads in flight
        variants = [("default (reader ring)", {}), ("default (reader ring)", {}), ("RG_READ_THREADS=4", {"RG_READ_THREADS": "4"}),
                    ("RG_READ_THREADS=16", {"RG_READ_THREADS": "16"}), ("RG_INGEST_MAP=1 (mapped, registered .bed)", {"RG_INGEST_MAP": "1"}),
                    ("RG_TIMING=1", {"RG_TIMING": "1"})]
    for name, env in variants:
        if name.startswith("sleep"):
            time.sleep(3)
        t0 = time.time()
        r = subprocess.run([exe, "--step", "1", "--bed", d + "/x", "--phenoFile", d + "/x.pheno", "--covarFile", d + "/x.covar", "--bsize", str(bs),
                            "--out", d + "/out"], capture_output=True, text=True, env=dict(os.environ, **env))
        dt = time.time() - t0
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        marks = [ln.strip() for ln in (r.stdout + r.stderr).split("\n") if "since start" in ln or "level 1 for" in ln or "complete (" in ln or "copied to the GPU" in ln or "[timing]" in ln]
        print("%-22s wall %.2f s = %.2e SNP*sample*pheno/s end to end | %s" % (name, dt, M * N * P / dt, " | ".join(marks)), flush=True)


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:]])
