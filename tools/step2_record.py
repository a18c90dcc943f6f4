#!/usr/bin/env python
"""Step-2 sub-record of bench.py's JSON line (BASELINE configs[4]'s shape: 500,000 samples, 10 phenotypes).

Per case one block of synthetic variants resident in host / device memory is scored through the C ABI (include/rg_step2.h) and the device
time between the library's own HIP events is reported:
  hard_calls                 2-bit rows (a .bed / hard-call .pgen block), no missing call          rg_s2_qt_block_packed
  hard_calls_missing_1pct    the same with 1 % missing calls (adds the missing-indicator contraction)
  hard_calls_masked_phenos   phenotypes that differ in their missing values (5 %): per-trait denominators
  dosages_8bit               8-bit BGEN-style dosages as integers in units of 1/255 (uint16 rows)  rg_s2_qt_block_int
  dosages_8bit_masked_phenos the same with masked phenotypes
Rooflines: the contraction kernels are bound by the i8 matrix cores on the op count the digit planes make (2 x 8 planes x (C + P)
columns x samples per variant and contracted set); next to it the rate at which the SOURCE encoding streams (bytes of the 2-bit rows or
of the 1-byte BGEN probabilities per second) as a fraction of the HBM peak -- SURVEY 8(f) calls the path "HBM/ingest-bound".
Parity: the first variants of every case against oracle/regenie_step2_qt.py (pinned against regenie's own Step-2 output) at the full
sample count.  cpu_baseline: that oracle (numpy, host cores) timed on the same variants -- kind "port"."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

PEAK_I8_TOPS = 5000.0     # as bench.py (2x the bf16 dense peak)
PEAK_HBM_GBS = 8000.0


def step2_record(n=500_000, C=10, P=10, bs_packed=8192, bs_int=1024, steps=6, warmup=2, n_check=24, torch=None):
    if torch is None:
        import torch
    from oracle import regenie_step2_qt as s2o        # checker / cpu_baseline only
    from regenie_amd.step2 import Step2QT
    rng = np.random.default_rng(1)
    X = np.linalg.qr(np.column_stack([np.ones(n), rng.normal(size=(n, C - 1))]))[0]
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(3)

    def null_model(masked):
        mask = (rng.random((n, P)) > 0.05) if masked else np.ones((n, P), bool)
        res = rng.normal(size=(n, P))
        res = (res - X @ (X.T @ res)) * mask
        res /= np.sqrt((res ** 2).sum(axis=0) / (mask.sum(axis=0) - C))
        return res, mask.astype(np.float64), np.ones(P)

    def gen_dosage(bs, miss_rate):
        maf = 0.05 + 0.45 * torch.rand(bs, 1, generator=g, device=dev)
        dd = (torch.rand(bs, n, generator=g, device=dev) < maf).to(torch.uint8) + (torch.rand(bs, n, generator=g, device=dev) < maf).to(torch.uint8)
        miss = (torch.rand(bs, n, generator=g, device=dev) < miss_rate) if miss_rate else None
        return dd, miss

    cases = {}
    cpu = None
    for name, kind, masked, miss_rate in (("hard_calls", "packed", False, 0.0), ("hard_calls_missing_1pct", "packed", False, 0.01),
                                          ("hard_calls_masked_phenos", "packed", True, 0.01), ("dosages_8bit", "int", False, 0.0),
                                          ("dosages_8bit_masked_phenos", "int", True, 0.0)):
        res, mask, scf = null_model(masked)
        bs = bs_packed if kind == "packed" else bs_int
        with Step2QT(n, C, P) as s2:
            s2.set_null(X.T, res.T, mask.T, scf)
            if kind == "packed":
                dd, miss = gen_dosage(bs, miss_rate)
                code = torch.where(dd == 2, torch.zeros_like(dd), torch.where(dd == 1, torch.full_like(dd, 2), torch.full_like(dd, 3)))
                if miss is not None:
                    code = torch.where(miss, torch.ones_like(code), code)
                c4 = code.view(bs, n // 4, 4)
                rows = (c4[:, :, 0] | (c4[:, :, 1] << 2) | (c4[:, :, 2] << 4) | (c4[:, :, 3] << 6)).contiguous()
                Gc = dd[:n_check].double()
                if miss is not None:
                    Gc[miss[:n_check]] = float("nan")
                Gc = Gc.cpu().numpy()
                del dd, code, c4, miss
                run = lambda: s2.score_block_packed(rows)                # noqa: E731
                src_bytes = bs * n / 4.0
                nset = 2 if miss_rate else 1
            else:    # 8-bit probabilities -> dosage in units of 1/255: P(het) + 2 P(hom) with two 8-bit numbers, 0 .. 510
                dd, _ = gen_dosage(bs, 0.0)
                noise = torch.randint(0, 40, (bs, n), generator=g, device=dev, dtype=torch.int32)
                d255 = torch.clamp(dd.to(torch.int32) * 255 - noise * (dd > 0) + noise * (dd == 0), 0, 510).to(torch.int32)
                G16 = d255.cpu().numpy().astype(np.uint16)
                Gc = G16[:n_check].astype(np.float64) / 255.0
                del dd, noise, d255
                run = lambda: s2.score_block_int(G16, 255)               # noqa: E731
                src_bytes = bs * n * 2.0                                   # two probability bytes per sample in the file
                nset = 1
            out = None
            ms = []
            for it in range(warmup + steps):
                out = run()
                if it >= warmup:
                    ms.append(out["kernel_ms"])
            t = float(np.mean(ms))
            t0 = time.perf_counter()
            want = s2o.score_qt_block_ref(Gc, X, res, mask, scf)
            t_or = time.perf_counter() - t0
            ok = np.isfinite(want["stats"])
            err = float(np.max(np.abs(out["stats"][:n_check][ok] - want["stats"][ok])) / np.max(np.abs(want["stats"][ok])))
            same_ign = bool(np.array_equal(out["ignored"][:n_check], want["ignored"]))
            ops = 2.0 * 8 * (C + P) * n * bs * nset
            cases[name] = {"ms_per_block": t, "block_variants": bs, "variants_per_s": bs / t * 1e3, "value": bs * n * P / t * 1e3,
                           "parity_vs_oracle": {"variants": n_check, "stats_max_rel_err": err, "ignored_flags_equal": same_ign},
                           "roofline_i8": {"bound": "mfma", "achieved": ops / t * 1e3 / 1e12, "peak": PEAK_I8_TOPS, "unit": "TOP/s", "frac": ops / t * 1e3 / 1e12 / PEAK_I8_TOPS,
                                           "algorithmic_ops_per_block": ops,
                                           "peak_note": "5,000 TOP/s = the dense i8 datasheet rate (2x bf16); the guide's MEASURED i8 ceiling is 3,944 TOP/s (frac x 1.27 against that)"},
                           "roofline_source_bytes": {"bound": "hbm", "achieved": src_bytes / t * 1e3 / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                                     "frac": src_bytes / t * 1e3 / 1e9 / PEAK_HBM_GBS, "bytes_per_block": src_bytes}}
            if name == "hard_calls_missing_1pct":
                cpu = {"value": n_check * n * P / t_or, "unit": "variant*sample*pheno/s", "cores": os.cpu_count(), "kind": "port",
                       "sample": "oracle/regenie_step2_qt.py (numpy, pinned to regenie's Step-2 output) on %d of the block's variants x %d samples x %d phenotypes, %.1f s" % (n_check, n, P, t_or)}
            if kind == "packed":
                del rows
            else:
                del G16
            torch.cuda.empty_cache()
    return {"metric": "Step-2 QT variants x samples x phenos / sec", "unit": "variant*sample*pheno/s", "value": cases["hard_calls_missing_1pct"]["value"],
            "dtype": "i8 digit planes (exact integer sums) + f64 recombination", "data": "synthetic",
            "config": {"workload": "BASELINE configs[4]'s shape on one GPU: %d samples, %d covariates, %d phenotypes; one block of variants per case, device time of the "
                                   "library's kernels (hard-call rows resident in HBM; integer dosages handed over as host rows, PCIe not in the figure)" % (n, C, P),
                       "samples": n, "covariates": C, "phenos": P},
            "cases": cases, "cpu_baseline": cpu,
            "note": "value = hard calls with 1 % missing calls (every block of array data has some)"}


if __name__ == "__main__":
    print(json.dumps(step2_record()))
