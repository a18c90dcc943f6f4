"""CPU restatement of regenie's Step-2 quantitative-trait score test (dense genotypes, default non-strict mode).

TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing in the product path).  numpy, fp64, the reference's own
operation order and orientation (samples down the rows).

PINNED against regenie itself: tests/test_reference_pin.py::test_step2_qt_oracle_against_reference compares BETA / SE /
CHISQ / LOG10P with the Step-2 output of oracle/_ref/regenie (the reference's sources compiled by oracle/Makefile) on
example_3chr.bed, fixtures under tests/golden/ref_outputs/step2/; closed-form identities in tests/test_step2_oracle.py.
"""
from __future__ import annotations

import math

import numpy as np

NUMTOL = 1e-6     # Regenie.hpp:220


def compute_res(Y, blup, masked_indivs, neff, ncov, scale_Y):
    """Data::compute_res, Data.cpp:2386-2400 (no --apply-rerint, blup not a covariate).
    Y, blup, masked_indivs: n x P.  Returns (res, p_sd_yres, scf_sv)."""
    res = (Y - blup) * masked_indivs                                   # :2390-2391
    p_sd_yres = np.sqrt((res ** 2).sum(axis=0)) / np.sqrt(neff - ncov)  # :2397-2398
    res = res / p_sd_yres                                              # :2399
    scf_sv = scale_Y * p_sd_yres                                       # :2400
    return res, p_sd_yres, scf_sv


def mean_impute(g):
    """Mean over the analysed non-missing samples, missing set to it (Geno.cpp:3183-3188; every row here is in the
    analysis).  Missing = NaN or the -3 code."""
    g = np.array(g, dtype=np.float64)
    miss = ~(g >= 0.0)
    nobs = int((~miss).sum())
    mean = g[~miss].sum() / nobs if nobs else float("nan")
    g[miss] = mean
    return g, mean, nobs


def residualize_geno(X, g, numtol=NUMTOL):
    """Geno.cpp:3242-3260.  X: n x C orthonormal.  Returns (scaled residual, scale_fac, ignored)."""
    beta = X.T @ g                      # :3246
    g = g - X @ beta                    # :3247
    sf = np.linalg.norm(g) / math.sqrt(X.shape[0] - X.shape[1])   # :3253-3254
    if not (sf >= numtol):              # :3256-3259
        return g, sf, True
    return g / sf, sf, False            # :3260


def score_qt(g_scaled, sf, res, masked_indivs, scf_sv):
    """compute_score_qt, dense non-strict branch, Step2_Models.cpp:346, 416-431 (flipped == false so gsc = scale_fac)."""
    gsc = sf
    num = (res.T @ g_scaled) * gsc                                     # :416
    denum = gsc * gsc * (masked_indivs.T.astype(np.float64) @ (g_scaled ** 2))   # :417
    stats = num / np.sqrt(denum)                                       # :420
    bhat = stats * scf_sv / np.sqrt(denum)                             # :427
    return stats, bhat


def score_qt_block(G, X, res, masked_indivs, scf_sv, numtol=NUMTOL):
    """compute_tests_mt over one block (Data.cpp:2476-2555).  G: bs x n raw genotypes / dosages with NaN or -3 missing.
    Returns a dict of arrays shaped like regenie_amd.step2.Step2QT.score_block's."""
    bs, P = G.shape[0], res.shape[1]
    out = {"stats": np.full((bs, P), np.nan), "bhat": np.full((bs, P), np.nan), "scale_fac": np.zeros(bs),
           "mean": np.zeros(bs), "n_obs": np.zeros(bs, np.int32), "ignored": np.zeros(bs, np.int32)}
    for j in range(bs):
        g, mean, nobs = mean_impute(G[j])
        out["mean"][j], out["n_obs"][j] = mean, nobs
        if nobs == 0:
            out["ignored"][j], out["scale_fac"][j] = 1, float("nan")
            continue
        gs, sf, ign = residualize_geno(X, g, numtol)
        out["scale_fac"][j], out["ignored"][j] = sf, int(ign)
        if ign:
            continue
        out["stats"][j], out["bhat"][j] = score_qt(gs, sf, res, masked_indivs, scf_sv)
    with np.errstate(invalid="ignore", divide="ignore"):
        out["se"] = out["bhat"] / out["stats"]        # :440
        out["chisq"] = out["stats"] ** 2              # :443
    return out


def get_logp(chisq):
    """-log10 p of a 1-df chi-square statistic, Regenie.cpp:1843-1856."""
    if chisq < 0 and abs(chisq) < 1e-6:
        return 0.0
    if chisq < 0:
        return -1.0
    pv = math.erfc(math.sqrt(chisq / 2.0))            # cdf(complement(chi_squared(1), T))
    if pv == 0:
        logp = math.log10(2) - 0.5 * math.log10(2 * math.pi * chisq) - 0.5 * chisq * math.log10(math.e)
    else:
        logp = math.log10(pv)
    return -logp
