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


def flip_geno(g_raw):
    """flip_geno (Geno.cpp:3150-3162), applied by regenie to every variant of an additive test on a binary or count trait (params.with_flip,
    Data.cpp:2108; never for quantitative traits): when the counted allele is the major one -- mean dosage of the analysed, observed samples
    above 1 -- the genotype becomes 2 - g (missing entries stay missing), so that the tests run on the MINOR allele; BETA is negated back
    afterwards (Step2_Models.cpp:621, :692).  The score test does not see it (sign aside); the corrections do: check_sparse_G and with it the
    fast forms of the saddlepoint approximation and of the approximate Firth fit (carriers only) look at the minor-allele coding.
    g_raw: dosages with negative values for missing (what decode_bed_rows returns, after --ref-first).  -> (g, flipped)"""
    obs = g_raw >= 0
    flipped = bool(obs.any() and g_raw[obs].mean() > 1.0)
    return (np.where(obs, 2.0 - g_raw, g_raw) if flipped else g_raw), flipped


def null_logistic(y_raw, X, mask, loco_offset, opt):
    """fit_null_logistic, test-mode branch (Step1_Models.cpp:54-140), for one phenotype: logistic regression of the trait on the
    covariate basis with the LOCO prediction as offset.  Returns None when it does not converge (the phenotype is skipped), else
    dict(p = Y_hat_p, gamma_sqrt = sqrt(p (1 - p)) (1 at masked samples), w)."""
    off = np.where(mask, loco_offset, 0.0)                              # :74 (the LOCO file holds NA for the samples masked for the trait)
    beta0 = np.zeros(X.shape[1])
    eta = off + X @ beta0
    p = orc.get_pvec(eta)                                               # :80
    ok, beta, p, eta = orc.fit_logistic(y_raw, X, off, mask, p, eta, beta0, opt, True, NUMTOL)
    if not ok:                                                          # (goes on from the state the first attempt left: the arguments are references)
        ok, beta, p, eta = orc.fit_logistic(y_raw, X, off, mask, p, eta, beta, opt, False, NUMTOL)
    if not ok:
        return None
    w, _ = orc.get_wvec(p, mask)                                        # :129
    return dict(p=p, w=w, gamma_sqrt=np.sqrt(w), beta=beta)             # :128-131


def score_bt(g, X, y_raw, mask, null, numtol=NUMTOL, sparse=False):
    """compute_score_bt for one mean-imputed variant and one phenotype (Step2_Models.cpp:486-520): GW = g Gamma_sqrt mask, projected off the
    orthonormal basis of Gamma X (getBasis of X_Gamma, Step1_Models.cpp:132-133), stats = Gres . yres / |Gres| with
    yres = (y - p) / Gamma_sqrt * mask (Data.cpp:2443-2445; NOT projected); get_sumstats (:2031-2041).

    sparse: check_sparse_G's verdict for the variant.  The sparse form takes GW . yres -- the genotype NOT projected -- as the numerator
    (:516-517) where the dense one takes Gres . yres (:519).  The two differ by (X^T W g)^T (X^T W X)^-1 X^T (y - p), the null model's
    score at ITS stopping point times the covariate coefficients of g: regenie stops that model at |score| < 1e-6, so the numbers part in
    the seventh digit -- which is what kept one line in ten of the product's uncorrected binary-trait rows from being byte-identical to
    regenie's until this was found (tests/golden/fuzz_driver_log.md; with it, 704 of 704 lines of the case it was found on)."""
    gs_mask = null["gamma_sqrt"] * mask
    XG, _ = orc.get_basis(X * gs_mask[:, None])
    GW = g * gs_mask
    Gres = GW - XG @ (XG.T @ GW)
    denum = float(Gres @ Gres)
    if np.sqrt(denum) < numtol:
        return None
    yres = (y_raw - null["p"]) / null["gamma_sqrt"] * mask
    stats = float((GW if sparse else Gres) @ yres) / np.sqrt(denum)
    se = 1.0 / np.sqrt(denum)
    return dict(stats=stats, se=se, bhat=stats * se, chisq=stats * stats, denum=denum, Gres=Gres)


def null_poisson(y_raw, X, mask, loco_offset, opt):
    """fit_null_poisson, test-mode branch (Step1_Models.cpp:225-288), for one count phenotype: Poisson regression on the covariate basis
    with the LOCO prediction as offset.  None when it does not converge, else dict(p = fitted rates, gamma_sqrt = sqrt(p), w = p)."""
    off = np.where(mask, loco_offset, 0.0)                              # :240
    p = y_raw + 1e-1                                                    # :244
    with np.errstate(invalid="ignore"):
        eta = np.where(mask, np.log(p), 0.0)                            # :245
    beta0 = np.zeros(X.shape[1])
    beta0[0] = eta.mean() - off.mean()                                  # :247
    ok, beta, p, eta = orc.fit_poisson(y_raw, X, off, mask, p, eta, beta0, opt)
    if not ok:
        return None
    return dict(p=p, w=p, gamma_sqrt=np.sqrt(p), beta=beta)             # :266-268


def score_ct(g, X, y_raw, mask, null, numtol=NUMTOL, sparse=False):
    """compute_score_ct (Step2_Models.cpp:559-622): as compute_score_bt with the Poisson weights; the variant is skipped for the trait
    when denum itself (not its root) is below numtol (:596).  sparse: as in score_bt (the sparse form's numerator is GW . yres, :602-603)."""
    gs_mask = null["gamma_sqrt"] * mask
    XG, _ = orc.get_basis(X * gs_mask[:, None])
    GW = g * gs_mask
    Gres = GW - XG @ (XG.T @ GW)
    denum = float(Gres @ Gres)
    if denum < numtol:
        return None
    yres = (y_raw - null["p"]) / null["gamma_sqrt"] * mask             # compute_res_count, Data.cpp:2457-2465
    stats = float((GW if sparse else Gres) @ yres) / np.sqrt(denum)
    se = 1.0 / np.sqrt(denum)
    return dict(stats=stats, se=se, bhat=stats * se, chisq=stats * stats)


# ---- approximate Firth correction (`--firth --approx`) ---------------------------------------------------------------------------------
# The reference reaches the maximisers below with a chain of solvers and fall-backs (fit_firth_nr, the pseudo-data IRLS of
# fit_firth_pseudo, step halving, restarts: Step2_Models.cpp:899-984, :1254-1737) that stop at |modified score| < 50 * numtol (null
# model) or < numtol_firth = 2.5e-4 (per variant).  The penalised likelihood is strictly concave in the range that matters, so its
# maximiser is unique: this restatement solves the same equations to machine precision with plain Fisher scoring + step halving, and
# agrees with regenie's printed numbers to its stopping tolerance (~1e-5 relative on BETA, less on CHISQ).  The per-variant fit is the
# exception since round 5: regenie's first solver there (the one-parameter fit_firth_pseudo) is restated to the letter, stopping rule
# included (pseudo_firth below), so that its rows come out in regenie's digits; the root finder stands in for the solvers behind it.

def _pvec(eta):
    return orc.get_pvec(eta)


def firth_null(y_raw, X, mask, offset, beta_start, maxit=2000, stop_tol=50 * NUMTOL):
    """fit_approx_firth_null / fit_firth_nr with every column free (Step2_Models.cpp:899-984, :1267-1385): maximise
    l(beta) + 0.5 log |X^T W X| over the covariate effects, the LOCO prediction as offset.  Modified score X^T (y - p + h (0.5 - p)),
    h = diag of the hat matrix of W^(1/2) X.  Returns beta (None if it does not converge).

    A sample that is masked for the trait (analysed, its phenotype missing for THIS trait) is not in the likelihood and not in the
    score -- but it IS in X^T W X, with weight 1: get_wvec returns `mask.select(p (1 - p), 1)` (Step1_Models.cpp:1809-1811) and
    fit_firth_nr builds X^T W X from that vector without applying the mask (:1287-1290, :1325-1328), so the penalty, the hat diagonal and
    the Newton matrix see the masked rows of X.  regenie's own single-trait run of the same trait (the samples dropped instead of masked)
    gives the masked-out value; its multi-trait run gives this one (1.2e-3 apart in the approximate Firth BETA of a drawn case with 16 of
    284 samples masked: tests/golden/fuzz_log.md).  The reference is the multi-trait program, so this follows it."""
    m = mask.astype(bool)
    Xm, ym, om = X[m], y_raw[m], np.nan_to_num(offset)[m]
    Xout = X[~m]                                       # rows that only enter X^T W X (weight 1)
    extra = Xout.T @ Xout

    def pen_dev(b):
        p = _pvec(om + Xm @ b)
        w = p * (1 - p)
        sign, logdet = np.linalg.slogdet(Xm.T @ (Xm * w[:, None]) + extra)
        return -2.0 * float(np.sum(np.where(ym == 0, np.log(1 - p), np.log(p)))) - logdet, p, w

    beta = np.array(beta_start, dtype=np.float64)
    dev, p, w = pen_dev(beta)
    for it in range(1, maxit + 1):
        XtWX = Xm.T @ (Xm * w[:, None]) + extra
        U = Xm * np.sqrt(w)[:, None]
        h = np.einsum("ij,ij->i", U @ np.linalg.inv(XtWX), U)
        score = Xm.T @ (ym - p + h * (0.5 - p))
        # fit_firth_nr's stopping rule (Step2_Models.cpp:1320-1323) with fit_approx_firth_null's tolerance 50 numtol = 5e-5 (:906): the estimates are those of
        # the iterate it stops at.  With masked samples in X^T W X the iteration is not Newton's any more and converges linearly, so that iterate lies a
        # tolerance away from the root -- and every approximate-Firth row of the trait inherits the difference in its last digits.
        if np.abs(score).max() < stop_tol and it >= 2:
            return beta
        if np.abs(score).max() < 1e-10:
            return beta
        step = np.linalg.solve(XtWX, score)
        if np.abs(step).max() < 1e-10:
            return beta
        mx = np.abs(step).max() / 25.0                                # maxstep_null
        if mx > 1:
            step = step / mx
        for _ in range(60):
            dev_new, p_new, w_new = pen_dev(beta + step)
            if dev_new < dev + 1e-12 or np.abs(step).max() < 1e-6:   # (steps that small change the deviance by less than its rounding)
                break
            step = step / 2
        beta, dev, p, w = beta + step, dev_new, p_new, w_new
    return None


NITER_FIRTH = 250            # params.niter_max_firth (Regenie.hpp:336)
TOL_FIRTH = 2.5e-4           # params.numtol_firth (:224)


def pseudo_firth(g, y, o, live, p0, x0, dev0, state, niter):
    """fit_firth_pseudo (Step2_Models.cpp:1548-1665), one parameter, started at 0.  g / y / o: the entries that enter (every sample with the
    masked ones at g = 0, or the carriers); state(b) -> (p, w, sum g^2 w, penalised deviance).  Returns (beta, se, lrt) or None (fit states 1 - 4)."""
    beta, beta14 = 0.0, 0.0
    gl = np.where(live, g, 0.0)
    for it in range(1, niter + 1):
        p, w, xtwx, dev = state(beta)
        h = gl * gl * w / xtwx
        ystar = y + h * (0.5 - p)
        score = float(np.sum(gl * (ystar - p)))
        if abs(score) < TOL_FIRTH and it >= 2:
            lrt = dev0 - dev
            return None if lrt < 0 else (beta, float(np.sqrt(1.0 / xtwx)), lrt)
        if it == 14:
            beta14 = beta
        if it == 15 and abs(beta - beta14) > 0.1:
            return None                                              # state 1: the Newton solver takes over
        bdiff, betanew = 1e16, beta
        for _ in range(25):                                          # unpenalised logistic regression on the pseudo-response
            step = score / xtwx
            bnew = abs(step)
            if bnew > bdiff:
                return None                                          # state 2
            mx = bnew / 5.0
            betanew = beta + (step / mx if mx > 1 else step)
            p = _pvec(o + g * betanew)
            score = float(np.sum(gl * (ystar - p)))
            if abs(score) < TOL_FIRTH:
                break
            w = np.where(live, p * (1 - p), 0.0)
            if (np.where(live, w, 1.0) == 0).any():
                return None                                          # state 3
            xtwx = float(np.sum(gl * gl * w))
            beta, bdiff = betanew, bnew
        beta = betanew
    return None                                                      # state 1: too slow


def firth_snp(y_raw, gvec, mask, offset, carriers=None, maxit=500, root=False):
    """fit_firth_logistic_snp_fast with its 1-parameter solvers (Step2_Models.cpp:1158-1253, :1548-1737): the variant's effect with
    the covariate effects of the null Firth model held in the offset; penalty 0.5 log(sum G^2 w), over the carriers only when
    `carriers` is given (the reference's fast approximation for sparse variants with MAC < 50, where the entries of G off the
    carriers are dropped from score and information as well).  Returns (beta, se, lrt) or None."""
    m = mask.astype(bool)

    def dev_all(b):
        p = _pvec(offset[m] + gvec[m] * b)
        return -2.0 * float(np.sum(np.where(y_raw[m] == 0, np.log(1 - p), np.log(p))))

    if carriers is not None:
        idx = np.asarray(carriers)
        g, y, o = gvec[idx], y_raw[idx], offset[idx]
        dev_non = dev_all(0.0) - (-2.0 * float(np.sum(np.where(y == 0, np.log(1 - _pvec(o)), np.log(_pvec(o))))))
    else:
        g, y, o = np.where(m, gvec, 0.0), y_raw, offset
        dev_non = 0.0
    live = m[idx] if carriers is not None else m

    def state(b):
        p = _pvec(o + g * b)
        w = np.where(live, p * (1 - p), 0.0) if carriers is None else p * (1 - p)
        xtwx = float(np.sum(g * g * w))
        ll = -2.0 * float(np.sum(np.where(live, np.where(y == 0, np.log(1 - p), np.log(p)), 0.0)))
        return p, w, xtwx, ll - np.log(xtwx)        # (the deviance of the samples left out is a constant: it cancels in the LRT)

    # dev0: the deviance of the offset-only model with the penalty of the SAME set of samples (:1206-1218)
    p0, w0, x0, dev0 = state(0.0)
    # regenie's first solver, to the letter: fit_firth_pseudo (:1548-1665) -- IRLS on the pseudo-response y* = y + h (0.5 - p), stopped at
    # |modified score| < numtol_firth = 2.5e-4, so its BETA / SE / LRT are those of the iterate it stops at, not of the root; only when it
    # gives up (slow, a growing step, p = 0, LRT < 0) do the Newton solvers run, which the root finder below stands in for
    fast = None if root else pseudo_firth(g, y, o, live, p0, x0, dev0, state, niter=NITER_FIRTH // 2 if carriers is not None else min(NITER_FIRTH // 2, 50))
    if fast is not None:                                              # (root=True: the maximiser itself, for tests of the equations)
        return fast
    beta = 0.0
    p, w, xtwx, dev = p0, w0, x0, dev0
    converged = False
    for _ in range(maxit):
        h = g * g * w / xtwx
        score = float(np.sum(np.where(live, g * (y + h * (0.5 - p) - p), 0.0)))
        step = score / xtwx
        if abs(step) < 1e-9:                                         # (below that the deviance comparisons of the step halving are rounding noise)
            converged = True
            break
        if abs(step) > 5:                                             # maxstep
            step = 5.0 * np.sign(step)
        ok = False
        for _ in range(40):
            p_n, w_n, x_n, dev_n = state(beta + step)
            if dev_n <= dev or abs(step) < 1e-6:                      # (steps that small change the deviance by less than its rounding)
                ok = True
                break
            step /= 2
        if not ok:                                                    # the deviance is flat to rounding: the root is reached
            converged = abs(score / xtwx) < 1e-6
            break
        beta, p, w, xtwx, dev = beta + step, p_n, w_n, x_n, dev_n
    if not converged:
        return None
    lrt = dev0 - dev
    if lrt < 0:
        return None
    return beta, float(np.sqrt(1.0 / xtwx)), lrt


def approx_firth(g, X, y_raw, mask, null, cov_blup_offset, sparse=False, mac=None):
    """The corrected statistic of one (variant, trait) whose score test exceeded the threshold: check_pval_snp -> run_firth_correction_snp
    -> fit_firth_logistic_snp_fast (Step2_Models.cpp:1987-2010, :2043-2066).  g: mean-imputed genotype; cov_blup_offset = X beta_null +
    LOCO prediction (fit_null_firth, :1011-1013).  Returns dict(bhat, se, chisq) or None (TEST_FAIL)."""
    gs_mask = null["gamma_sqrt"] * mask
    XG, _ = orc.get_basis(X * gs_mask[:, None])
    GW = g * gs_mask
    Gres = GW - XG @ (XG.T @ GW)                                        # :528-531 / :503
    gvec = Gres / null["gamma_sqrt"]                                    # :2061
    carriers = None
    if sparse and mac is not None and mac < 50:                        # :1174-1185
        carriers = np.flatnonzero((mask > 0) & (g > 1e-4))
    out = firth_snp(y_raw, gvec, mask, cov_blup_offset, carriers)
    if out is None:
        return None
    beta, se, lrt = out
    return dict(bhat=beta, se=se, chisq=lrt)


def firth_fit(y_raw, X, mask, offset, beta_start, nfree, maxstep=25.0, maxit=2000):
    """fit_firth_nr with cols_incl = nfree (Step2_Models.cpp:1267-1385): maximise l(beta) + 0.5 log |X^T W X| over the FIRST nfree
    coefficients, the others held at their start values; the penalty and the hat diagonal always use every column of X.  Returns
    (beta, penalised deviance, (X^T W X)^-1) or None."""
    m = mask.astype(bool)
    Xm, ym, om = X[m], y_raw[m], offset[m]
    extra = X[~m].T @ X[~m]                            # the masked rows enter X^T W X with weight 1 (see firth_null)

    def pen_dev(b):
        p = _pvec(om + Xm @ b)
        w = p * (1 - p)
        sign, logdet = np.linalg.slogdet(Xm.T @ (Xm * w[:, None]) + extra)
        return -2.0 * float(np.sum(np.where(ym == 0, np.log(1 - p), np.log(p)))) - logdet, p, w

    beta = np.array(beta_start, dtype=np.float64)
    dev, p, w = pen_dev(beta)
    for _ in range(maxit):
        XtWX = Xm.T @ (Xm * w[:, None]) + extra
        inv = np.linalg.inv(XtWX)
        U = Xm * np.sqrt(w)[:, None]
        h = np.einsum("ij,ij->i", U @ inv, U)
        score = Xm[:, :nfree].T @ (ym - p + h * (0.5 - p))
        step = np.zeros_like(beta)
        step[:nfree] = np.linalg.solve(XtWX[:nfree, :nfree], score)
        if np.abs(step).max() < 1e-10:
            return beta, dev, inv
        mx = np.abs(step).max() / maxstep
        if mx > 1:
            step = step / mx
        for _ in range(60):
            dev_new, p_new, w_new = pen_dev(beta + step)
            if dev_new < dev + 1e-12 or np.abs(step).max() < 1e-6:
                break
            step = step / 2
        beta, dev, p, w = beta + step, dev_new, p_new, w_new
    return None


def exact_firth(g, X, y_raw, mask, loco_offset, beta_cov_start):
    """The exact Firth test of one (variant, trait) (`--firth` without `--approx`): fit_firth_logistic_snp, null fit then full fit
    (Step2_Models.cpp:1062-1156; run_firth_correction_snp :2045-2051).  Design = [covariates | g] with g the mean-imputed genotype on its raw
    scale; null = the maximiser with the variant's coefficient held at 0 UNDER THE SAME PENALTY; LRT = difference of the penalised deviances;
    SE from (X^T W X)^-1 at the full maximiser."""
    C = X.shape[1]
    Xf = np.column_stack([X, g])
    off = np.nan_to_num(loco_offset)
    nul = firth_fit(y_raw, Xf, mask, off, np.concatenate([beta_cov_start, [0.0]]), C)
    if nul is None:
        return None
    full = firth_fit(y_raw, Xf, mask, off, nul[0], C + 1, maxstep=5.0)
    if full is None:
        return None
    lrt = nul[1] - full[1]
    if lrt < 0:
        return None
    return dict(bhat=float(full[0][C]), se=float(np.sqrt(full[2][C, C])), chisq=lrt)


# ---- saddlepoint approximation (`--spa`) -----------------------------------------------------------------------------------------------
MAX_EXP_LIM = 708.0          # Step2_Models.hpp
TOL_SPA = np.finfo(np.float64).eps ** 0.25      # params.tol_spa (Regenie.hpp:330): regenie stops Newton at |K'(t) - s| < 1.2e-4, so the
NITER_SPA = 1000                                 # iteration itself is restated (an exact root would differ in the 4th digit)


def spa_test(stats, denum, Gres, null, mask, carriers=None):
    """run_SPA_test_snp and its helpers (Step2_Models.cpp:2072-2297): the two-sided p-value of the score statistic from the saddlepoint
    approximation of its null distribution (Lugannani-Rice), cumulant generating function K of sum_i Gmod_i (Y_i - p_i) / c with
    Gmod = Gres / Gamma_sqrt.  carriers (the non-zero entries of the mean-imputed genotype, unmasked): regenie's fast form for sparse
    variants -- exact terms for the carriers, a normal approximation for everyone else.  Returns dict(chisq, logp, bhat, se) or None."""
    from scipy.stats import norm
    m = mask.astype(bool)
    phat, gam = null["p"], null["gamma_sqrt"]
    c = np.sqrt(denum)
    Gmod = np.where(m, Gres / gam, 0.0)
    Gmu = Gmod * phat
    a = float(Gmu[m].sum())
    fast = carriers is not None
    if fast:
        idx = np.asarray([j for j in carriers if m[j]], dtype=np.int64)
        b = denum - float((Gres[idx] ** 2).sum())
        d = float(Gmu[idx].sum())
        gm, ph, gs = Gmod[idx], phat[idx], gam[idx]
    else:
        gm, ph, gs = Gmod[m], phat[m], gam[m]
    score_num = stats * c
    lo = float(Gmod[m][Gmod[m] < 0].sum()) - a
    hi = float(Gmod[m][Gmod[m] > 0].sum()) - a
    if score_num < lo or score_num > hi:
        return None

    def K(t):
        v = float(np.log(1 - ph + ph * np.exp(t / c * gm)).sum())
        return v + (-t * d / c + t * t / 2 / denum * b if fast else -t * a / c)

    def K1(t):
        v = float(((gm * ph / c) / (ph + (1 - ph) * np.exp(-t / c * gm))).sum())
        return v + (-d / c + t / denum * b if fast else -a / c)

    def K2(t):
        vexp = -t / c * gm
        if (vexp > MAX_EXP_LIM).any():
            return 0.0
        v = float(((gm * gm * gs * gs / (c * c) * np.exp(vexp)) / (ph + (1 - ph) * np.exp(vexp)) ** 2).sum())
        return v + (b / denum if fast else 0.0)

    tval = -abs(stats)

    def solve(lam):                                                    # solve_K1_snp: Newton with a bisection safeguard
        min_x, max_x = (0.0, np.finfo(np.float64).max) if tval >= 0 else (np.finfo(np.float64).min, 0.0)
        t_old = 0.0
        f_old = lam * K1(lam * t_old) - tval
        t_new = -1.0
        for _ in range(NITER_SPA):
            hess = K2(lam * t_old)
            if hess == 0:
                return None
            t_new = t_old - f_old / hess
            f_new = lam * K1(lam * t_new) - tval
            if abs(f_new) < TOL_SPA:
                return t_new
            if t_new and min_x < t_new < max_x:
                if f_new > 0:
                    max_x = t_new
                else:
                    min_x = t_new
            else:
                t_new = (min_x + max_x) / 2
                f_new = lam * K1(lam * t_new) - tval
                if f_new <= 0:
                    min_x = t_new
                else:
                    max_x = t_new
            t_old, f_old = t_new, f_new
        return None

    pv = 0.0
    for lam in (1, -1):
        root = solve(lam)
        if root is None:
            return None
        kval, k2val = K(lam * root), K2(lam * root)
        if k2val == 0:
            return None
        wval = np.sign(root) * np.sqrt(2 * (root * tval - kval))
        vval = root * np.sqrt(k2val)
        pv += 0.5 if vval == 0 else float(norm.cdf(wval + np.log(vval / wval) / wval))
    if pv > 1:
        return None
    pval = max(10.0 * np.finfo(np.float64).tiny, pv)                   # get_logp(pv, ...), Regenie.cpp:1859-1873
    chisq = float(norm.isf(pval / 2) ** 2)
    se = 1.0 / np.sqrt(denum)                                          # check_pval_snp :2019-2020
    return dict(chisq=chisq, logp=-np.log10(pval), se=se, bhat=np.sign(stats) * np.sqrt(chisq) * se)
