#!/usr/bin/env python
"""One JSON line for the Step-2 QT hard-call route in the shape of bench.py's: rows resident in HBM, 200,000 samples
(round 2's size; BASELINE configs[4] has 500,000 samples -- tools/step2_record.py measures that), C = 10 covariate basis columns, P = 10 phenotypes, blocks of 16,384 variants, device time of the
library's kernels between its own HIP events.  Roofline: the contraction kernel k_xy_i8 is bound by the i8 MFMA pipe -- algorithmic work
= 2 x 8 digit planes x (C + P) columns x samples integer operations per variant and contracted set (the missing-indicator set doubles it
for blocks with missing calls); peak from /opt/skills/guides/MI355X_MICROARCH.md (I8 dense >= 3944 TOPS).  Usage (GPU box):
python tools/step2_bench_line.py > gpurun_out/<dir>/r2_step2_bench_line.json"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from regenie_amd.step2 import Step2QT


def main(n=200_000, C=10, P=10, bs=16384, steps=8, warmup=2):
    rng = np.random.default_rng(1)
    X = np.linalg.qr(np.column_stack([np.ones(n), rng.normal(size=(n, C - 1))]))[0]
    res = rng.normal(size=(n, P))
    res -= X @ (X.T @ res)
    res /= res.std(axis=0)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(3)
    out = {}
    with Step2QT(n, C, P) as s2:
        s2.set_null(X.T, res.T, np.ones((P, n), np.uint8), np.ones(P))
        for name, miss_rate in (("no_missing_call", 0.0), ("missing_calls_1pct", 0.01)):
            maf = 0.05 + 0.45 * torch.rand(bs, 1, generator=g, device=dev)
            dd = (torch.rand(bs, n, generator=g, device=dev) < maf).to(torch.uint8) + (torch.rand(bs, n, generator=g, device=dev) < maf).to(torch.uint8)
            code = torch.where(dd == 2, torch.zeros_like(dd), torch.where(dd == 1, torch.full_like(dd, 2), torch.full_like(dd, 3)))
            if miss_rate:
                code = torch.where(torch.rand(bs, n, generator=g, device=dev) < miss_rate, torch.ones_like(code), code)
            c = code.view(bs, n // 4, 4)
            rows = (c[:, :, 0] | (c[:, :, 1] << 2) | (c[:, :, 2] << 4) | (c[:, :, 3] << 6)).contiguous()
            del dd, code, c
            ms = [s2.score_block_packed(rows)["kernel_ms"] for _ in range(warmup + steps)][warmup:]
            t = float(np.mean(ms))
            nset = 2 if miss_rate else 1
            ops = 2.0 * 8 * (C + P) * n * bs * nset
            out[name] = {"ms_per_block": t, "variants_per_s": bs / t * 1e3, "value": bs * n * P / t * 1e3,
                         "roofline": {"kernel": "k_xy_i8 (+ k_s2_rows, k_s2_combine, k_s2_packed_final in the same timed region)", "bound": "mfma",
                                      "achieved": ops / t * 1e3 / 1e12, "peak": 3944.0, "unit": "TOP/s (int8)", "frac": ops / t * 1e3 / 1e12 / 3944.0,
                                      "algorithmic_ops_per_block": ops, "hbm_GBps_of_2bit_rows": bs * n / 4 / t * 1e3 / 1e9}}
            del rows
    line = {"metric": "Step-2 QT variants x samples x phenos / sec (hard calls, rg_s2_qt_block_packed)", "unit": "variant*sample*pheno/s",
            "value": out["missing_calls_1pct"]["value"], "n_gpus": 1, "steps": steps, "warmup": warmup, "dtype": "i8 digit planes (exact integer sums) + f64 recombination",
            "data": "synthetic", "higher_is_better": True,
            "config": {"workload": "200000 samples (BASELINE configs[4] itself has 500000: see tools/step2_record.py), 10 covariates, 10 phenotypes, blocks of 16384 variants, rows resident in HBM",
                       "samples": n, "covariates": C, "phenos": P, "block": bs},
            "cases": out,
            "note": "value = the case with 1 % missing calls (every block of real array data has some, which adds the missing-indicator contraction); "
                    "end to end from files: profiles/r2_step2_e2e.md (0.62 s against regenie's 5.3 s at 50k x 100k)"}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
