#!/usr/bin/env python
"""Step-2 QT, hard calls: device time of rg_s2_qt_block_packed against rg_s2_qt_block (fp64 rows) at 200,000 samples (BASELINE configs[4] has 500,000; tools/step2_record.py), the round-2 probe's sample
count (200,000 samples, 10 covariates, 10 phenotypes), for several block sizes.  Usage (GPU box): python tools/step2_packed_probe.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from regenie_amd.step2 import Step2QT


def main(n=200_000, C=10, P=10):
    rng = np.random.default_rng(1)
    X = np.linalg.qr(np.column_stack([np.ones(n), rng.normal(size=(n, C - 1))]))[0]
    res = rng.normal(size=(n, P))
    res -= X @ (X.T @ res)
    res /= res.std(axis=0)
    mask = np.ones((n, P), np.uint8)
    scf = np.ones(P)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(3)
    with Step2QT(n, C, P) as s2:
        s2.set_null(X.T, res.T, mask.T, scf)
        for bs in (512, 1024, 4096, 16384):
            rows = torch.randint(0, 256, (bs, (n + 3) // 4), dtype=torch.uint8, device=dev, generator=g)   # uniform codes: 25% missing
            rows_nomiss = rows | ((rows & 0x55) & ~((rows >> 1) & 0x55)) << 1                                    # 01 -> 11
            for name, rr in (("25% missing calls", rows), ("no missing call", rows_nomiss.to(torch.uint8))):
                ms = []
                for _ in range(5):
                    out = s2.score_block_packed(rr)
                    ms.append(out["kernel_ms"])
                t = float(np.median(ms[1:]))
                print("packed  bs %6d  %-18s %8.3f ms  %7.2f M variants/s  %6.1f GB/s of 2-bit rows  %.2e genotype*pheno/s"
                      % (bs, name, t, bs / t / 1e3, bs * n / 4 / t / 1e6, bs * n * P / t * 1e3), flush=True)
            if bs <= 4096 and "--no-fp64" not in sys.argv:
                G = torch.randint(0, 3, (bs, n), device=dev, generator=g).double()
                ms = [s2.score_block(G)["kernel_ms"] for _ in range(4)]
                t = float(np.median(ms[1:]))
                print("fp64    bs %6d  %-18s %8.3f ms  %7.2f M variants/s  %6.1f GB/s of fp64 rows" % (bs, "", t, bs / t / 1e3, 2 * bs * n * 8 / t / 1e6), flush=True)
                del G
    # phenotypes that differ in their missing values (5 %): C * P + P more columns (136 instead of 20), the g^2 contraction, both branches
    mask2 = (rng.random((n, P)) > 0.05).astype(np.uint8)
    with Step2QT(n, C, P) as s2:
        s2.set_null(X.T, (res * mask2).T, mask2.T, scf)
        for bs in (1024, 4096):
            rows = torch.randint(0, 256, (bs, (n + 3) // 4), dtype=torch.uint8, device=dev, generator=g)
            rr = (rows | ((rows & 0x55) & ~((rows >> 1) & 0x55)) << 1).to(torch.uint8)
            ms = [s2.score_block_packed(rr)["kernel_ms"] for _ in range(4)]
            t = float(np.median(ms[1:]))
            print("packed, masked phenotypes  bs %6d  %8.3f ms  %7.2f M variants/s  (136 columns)" % (bs, t, bs / t / 1e3), flush=True)
        # the contraction primitive with the binary-trait test's columns: P * (C + 3) = 130, the first P also against g^2
        cols = rng.normal(size=(P * (C + 3), n))
        s2.set_columns(cols, n_sq=P)
        for bs in (1024, 4096):
            rows = torch.randint(0, 256, (bs, (n + 3) // 4), dtype=torch.uint8, device=dev, generator=g)
            rr = (rows | ((rows & 0x55) & ~((rows >> 1) & 0x55)) << 1).to(torch.uint8).cpu().numpy()
            ms = [s2.contract_packed(rr)["kernel_ms"] for _ in range(4)]
            t = float(np.median(ms[1:]))
            print("contraction primitive (bt)  bs %6d  %8.3f ms  %7.2f M variants/s  (130 columns)" % (bs, t, bs / t / 1e3), flush=True)
    # integer dosages (8-bit .bgen probabilities as uint16 in units of 1 / 255): two digit planes + the missing indicator
    for msk, name in ((np.ones((n, P), np.uint8), "complete phenotypes"), (mask2, "masked phenotypes (lists)")):
        with Step2QT(n, C, P) as s2:
            s2.set_null(X.T, (res * msk).T, msk.T, scf)
            for bs in (1024, 4096):
                hard = rng.binomial(2, 0.2, size=(bs, n))
                Gi = np.clip(hard * 255 + (rng.random((bs, n)) < 0.4) * rng.integers(-80, 81, size=(bs, n)), 0, 510).astype(np.uint16)
                ms = [s2.score_block_int(Gi, 255)["kernel_ms"] for _ in range(4)]
                t = float(np.median(ms[1:]))
                fp = s2.score_block(Gi[:512].astype(np.float64) / 255.0)["kernel_ms"] if bs == 1024 else None
                print("integer dosages (scale 255), %-26s bs %6d  %8.3f ms  %7.2f M variants/s%s"
                      % (name, bs, t, bs / t / 1e3, "" if fp is None else "   (fp64 route: %.3f ms per 512)" % fp), flush=True)


if __name__ == "__main__":
    main()
