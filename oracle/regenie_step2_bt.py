"""CPU restatement of regenie's Step-2 binary-trait score test without the Firth / SPA corrections (`--step 2 --bt`) and of the
count-trait score test (`--step 2 --ct`).

TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing in the product path).  numpy, fp64, samples down the rows.

PINNED against regenie itself: tests/test_reference_pin.py::test_step2_bt_oracle_against_reference compares BETA / SE / CHISQ /
LOG10P / A1FREQ / N with the output of oracle/_ref/regenie for `--step 2 --bt --bed example --remove ...` fed by the reference's own
Step-1 LOCO files (tests/golden/ref_outputs/step2/bt_score_bed_Y*.regenie.gz); ::test_step2_ct_oracle_against_reference does the same for
count traits on synthetic data (ct_synth: regenie's own --step 1 --ct and --step 2 --ct outputs).
"""
from __future__ import annotations

import numpy as np

from . import regenie_step1 as orc

NUMTOL = 1e-6


def null_logistic(y_raw, X, mask, loco_offset, opt):
    """fit_null_logistic, test-mode branch (Step1_Models.cpp:54-140), for one phenotype: logistic regression of the trait on the
    covariate basis with the LOCO prediction as offset.  Returns None when it does not converge (the phenotype is skipped), else
    dict(p = Y_hat_p, gamma_sqrt = sqrt(p (1 - p)) (1 at masked samples), w)."""
    off = loco_offset * mask                                            # :74
    beta0 = np.zeros(X.shape[1])
    eta = off + X @ beta0
    p = orc.get_pvec(eta)                                               # :80
    ok, beta, p, eta = orc.fit_logistic(y_raw, X, off, mask, p, eta, beta0, opt, True, NUMTOL)
    if not ok:
        ok, beta, p, eta = orc.fit_logistic(y_raw, X, off, mask, p, eta, beta0, opt, False, NUMTOL)
    if not ok:
        return None
    w, _ = orc.get_wvec(p, mask)                                        # :129
    return dict(p=p, w=w, gamma_sqrt=np.sqrt(w), beta=beta)             # :128-131


def score_bt(g, X, y_raw, mask, null, numtol=NUMTOL):
    """compute_score_bt for one mean-imputed variant and one phenotype (Step2_Models.cpp:486-520, dense form; the sparse form is the
    same number): GW = g Gamma_sqrt mask, projected off the orthonormal basis of Gamma X (getBasis of X_Gamma, Step1_Models.cpp:132-133),
    stats = Gres . yres / |Gres| with yres = (y - p) / Gamma_sqrt * mask (Data.cpp:2443-2445); get_sumstats (:2031-2041)."""
    gs_mask = null["gamma_sqrt"] * mask
    XG, _ = orc.get_basis(X * gs_mask[:, None])
    GW = g * gs_mask
    Gres = GW - XG @ (XG.T @ GW)
    denum = float(Gres @ Gres)
    if np.sqrt(denum) < numtol:
        return None
    yres = (y_raw - null["p"]) / null["gamma_sqrt"] * mask
    stats = float(Gres @ yres) / np.sqrt(denum)
    se = 1.0 / np.sqrt(denum)
    return dict(stats=stats, se=se, bhat=stats * se, chisq=stats * stats)


def null_poisson(y_raw, X, mask, loco_offset, opt):
    """fit_null_poisson, test-mode branch (Step1_Models.cpp:225-288), for one count phenotype: Poisson regression on the covariate basis
    with the LOCO prediction as offset.  None when it does not converge, else dict(p = fitted rates, gamma_sqrt = sqrt(p), w = p)."""
    off = loco_offset * mask                                            # :240
    p = y_raw + 1e-1                                                    # :244
    with np.errstate(invalid="ignore"):
        eta = np.where(mask, np.log(p), 0.0)                            # :245
    beta0 = np.zeros(X.shape[1])
    beta0[0] = eta.mean() - off.mean()                                  # :247
    ok, beta, p, eta = orc.fit_poisson(y_raw, X, off, mask, p, eta, beta0, opt)
    if not ok:
        return None
    return dict(p=p, w=p, gamma_sqrt=np.sqrt(p), beta=beta)             # :266-268


def score_ct(g, X, y_raw, mask, null, numtol=NUMTOL):
    """compute_score_ct (Step2_Models.cpp:559-622): as compute_score_bt with the Poisson weights; the variant is skipped for the trait
    when denum itself (not its root) is below numtol (:596)."""
    gs_mask = null["gamma_sqrt"] * mask
    XG, _ = orc.get_basis(X * gs_mask[:, None])
    GW = g * gs_mask
    Gres = GW - XG @ (XG.T @ GW)
    denum = float(Gres @ Gres)
    if denum < numtol:
        return None
    yres = (y_raw - null["p"]) / null["gamma_sqrt"] * mask             # compute_res_count, Data.cpp:2457-2465
    stats = float(Gres @ yres) / np.sqrt(denum)
    se = 1.0 / np.sqrt(denum)
    return dict(stats=stats, se=se, bhat=stats * se, chisq=stats * stats)
