"""CPU restatement of regenie's Step-2 quantitative-trait score test (dense genotypes, default non-strict mode).

TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing in the product path).  numpy, fp64, the reference's own
operation order and orientation (samples down the rows).

PINNED against regenie itself: tests/test_reference_pin.py::test_step2_qt_oracle_against_reference compares BETA / SE /
CHISQ / LOG10P with the Step-2 output of oracle/_ref/regenie (the reference's sources compiled by oracle/Makefile) on
example_3chr.bed, and ::test_step2_qt_oracle_sparse_branch_against_reference pins score_qt_block_ref (the sparse-genotype branch
the reference takes for variants with at most half of the samples non-zero, which differs from the dense one when phenotypes
differ in their missing values) on synthetic data with 5 % missing phenotypes; fixtures under tests/golden/ref_outputs/step2/;
closed-form identities in tests/test_step2_oracle.py.
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


def check_sparse(g_imputed, n_samples, prop_zero_thr=0.5):
    """check_sparse_G for .bed input (n_zero == -1), Geno.cpp:3165-3177: after the mean imputation, at most
    n_samples * (1 - prop_zero_thr) analysed entries are non-zero.  n_samples = params.n_samples (every kept sample of the file,
    analysed or not)."""
    return int(np.count_nonzero(g_imputed)) <= n_samples * (1.0 - prop_zero_thr)


def score_qt_sparse(g, X, res, masked_indivs, scf_sv, YtX):
    """compute_score_qt, the sparse non-strict branch, Step2_Models.cpp:402-413, 420-427: the mean-imputed genotype on its raw
    scale, covariates projected out of the numerator through YtX = res^T X (Data.cpp:2402), and per trait
    denum = |G m|^2 - 2 (X^T (G m)) . (X^T G) + |X^T G|^2 -- "an approximation assuming X'X is same for all traits (=I)"."""
    XtG = X.T @ g                                                       # :403
    num = res.T @ g - YtX @ XtG                                         # :404
    XtG_ss = float(XtG @ XtG)                                           # :405
    P = res.shape[1]
    denum = np.empty(P)
    for ph in range(P):                                                 # :407-410
        gm = g * masked_indivs[:, ph]
        XtGm = X.T @ gm
        denum[ph] = float(gm @ gm) - 2.0 * float(XtGm @ XtG) + XtG_ss
    with np.errstate(invalid="ignore", divide="ignore"):               # an all-zero variant: 0 / 0 (the MAC filter drops it earlier)
        stats = num / np.sqrt(denum)                                    # :420
        bhat = stats * scf_sv / np.sqrt(denum)                          # :427
    return stats, bhat


def score_qt_block_ref(G, X, res, masked_indivs, scf_sv, n_samples=None, numtol=NUMTOL, prop_zero_thr=0.5, zero_count_rule=False):
    """compute_tests_mt over one block as the reference runs it for hard calls (Data.cpp:2476-2555): check_sparse_G decides per
    variant between the sparse branch (no residualisation, scale_fac = 1: Data.cpp:2513-2515) and the dense one.  With every
    mask entry 1 the two branches are the same number; they differ when phenotypes differ in their missing values.
    zero_count_rule: check_sparse_G's OTHER form, the one .pgen input takes (readChunkFromPGENFileToG counts the exact zeros among the
    analysed, observed samples while it parses -- Geno.cpp:2581, :2594; its start value, ind_in_analysis.size() - n_samples, is zero -- then
    is_sparse = n_zero >= n_samples * prop_zero_thr, :3171): a missing call is not a zero there, where the .bed / .bgen form counts the
    non-zeros AFTER the mean imputation.  Extra output: "sparse" [bs]."""
    bs, P = G.shape[0], res.shape[1]
    n_samples = X.shape[0] if n_samples is None else n_samples
    YtX = res.T @ X                                                    # Data.cpp:2402
    out = {"stats": np.full((bs, P), np.nan), "bhat": np.full((bs, P), np.nan), "scale_fac": np.zeros(bs),
           "mean": np.zeros(bs), "n_obs": np.zeros(bs, np.int32), "ignored": np.zeros(bs, np.int32), "sparse": np.zeros(bs, np.int32)}
    for j in range(bs):
        g, mean, nobs = mean_impute(G[j])
        out["mean"][j], out["n_obs"][j] = mean, nobs
        if nobs == 0:
            out["ignored"][j], out["scale_fac"][j] = 1, float("nan")
            continue
        sparse = int(np.count_nonzero(G[j] == 0.0)) >= n_samples * prop_zero_thr if zero_count_rule else check_sparse(g, n_samples, prop_zero_thr)
        if sparse:
            out["sparse"][j], out["scale_fac"][j] = 1, 1.0
            out["stats"][j], out["bhat"][j] = score_qt_sparse(g, X, res, masked_indivs, scf_sv, YtX)
            continue
        gs, sf, ign = residualize_geno(X, g, numtol)
        out["scale_fac"][j], out["ignored"][j] = sf, int(ign)
        if ign:
            continue
        out["stats"][j], out["bhat"][j] = score_qt(gs, sf, res, masked_indivs, scf_sv)
    with np.errstate(invalid="ignore", divide="ignore"):
        out["se"] = out["bhat"] / out["stats"]
        out["chisq"] = out["stats"] ** 2
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
