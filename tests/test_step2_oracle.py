"""The Step-2 QT oracle (oracle/regenie_step2_qt.py) against closed-form identities -- the reference has no QT Step-2
golden output to pin it on (see the oracle's header)."""
import numpy as np
import pytest
from scipy import stats as sps

from oracle import regenie_step2_qt as s2


def _problem(seed, n=400, C=4, P=3, bs=12, miss_y=True):
    rng = np.random.default_rng(seed)
    cov = np.column_stack([np.ones(n), rng.normal(size=(n, C - 1))])
    X = np.linalg.qr(cov)[0]                                   # new_cov: orthonormal basis incl. the intercept
    Y = rng.normal(size=(n, P))
    mask = np.ones((n, P))
    if miss_y:
        mask[rng.random((n, P)) < 0.05] = 0
    Y = (Y - X @ (X.T @ Y)) * mask
    neff = mask.sum(axis=0)
    scale_Y = np.sqrt((Y ** 2).sum(axis=0) / (neff - C))
    Y = Y / scale_Y
    blup = 0.1 * rng.normal(size=(n, P)) * mask
    res, p_sd, scf = s2.compute_res(Y, blup, mask, neff, C, scale_Y)
    G = rng.binomial(2, rng.uniform(0.05, 0.5, size=bs)[:, None], size=(bs, n)).astype(np.float64)
    return X, Y, blup, mask, res, scf, scale_Y, G


def test_frisch_waugh_and_partial_correlation():
    """No phenotype missingness: bhat is the OLS coefficient of the original-scale residual phenotype on g adjusted for
    the covariates, and stats = sqrt(n - C) * partial correlation."""
    X, Y, blup, mask, res, scf, scale_Y, G = _problem(1, miss_y=False)
    n, C = X.shape
    out = s2.score_qt_block(G, X, res, mask, scf)
    y_orig = (Y - blup) * scale_Y            # what res * scf_sv reconstructs
    for j in range(G.shape[0]):
        for p in range(Y.shape[1]):
            coef = np.linalg.lstsq(np.column_stack([X, G[j]]), y_orig[:, p], rcond=None)[0][-1]
            assert out["bhat"][j, p] == pytest.approx(coef, rel=1e-9, abs=1e-12)
            r = G[j] - X @ (X.T @ G[j])
            yr = res[:, p] - X @ (X.T @ res[:, p])
            # res is not re-residualised by the reference; the statistic uses res . r = yr . r
            rho = (yr @ r) / (np.linalg.norm(res[:, p]) * np.linalg.norm(r))
            assert out["stats"][j, p] == pytest.approx(rho * np.sqrt(n - C), rel=1e-9)
    assert np.allclose(out["se"], out["bhat"] / out["stats"])


def test_missing_genotypes_and_ignored_variants():
    X, Y, blup, mask, res, scf, scale_Y, G = _problem(2)
    G[0, ::7] = np.nan
    G[1, ::5] = -3.0
    G[2, :] = 1.0                  # monomorphic: residual is zero -> ignored
    G[3, :] = np.nan               # nothing observed
    out = s2.score_qt_block(G, X, res, mask, scf)
    assert out["n_obs"][0] == G.shape[1] - len(range(0, G.shape[1], 7))
    assert out["ignored"].tolist()[:4] == [0, 0, 1, 1]
    assert np.isnan(out["stats"][2]).all() and np.isnan(out["stats"][3]).all()
    g0 = G[0].copy(); obs = ~np.isnan(g0); g0[~obs] = g0[obs].mean()
    ref = s2.score_qt_block(g0[None], X, res, mask, scf)
    assert np.array_equal(ref["stats"][0], out["stats"][0])
    # per-phenotype denominators respect the phenotype's own missingness
    r = g0 - X @ (X.T @ g0)
    for p in range(Y.shape[1]):
        assert out["stats"][0, p] == pytest.approx((res[:, p] @ r) / np.sqrt((mask[:, p] * r * r).sum()), rel=1e-10)


def test_get_logp():
    for t in (0.0, 0.5, 3.84, 30.0, 200.0, 1400.0):
        assert s2.get_logp(t) == pytest.approx(-sps.chi2.logsf(t, 1) / np.log(10), rel=1e-6, abs=1e-12)
    assert s2.get_logp(5000.0) > 1000          # the asymptotic branch (pv underflows)
    assert s2.get_logp(-1e-9) == 0.0 and s2.get_logp(-1.0) == -1.0


# ---- the branches and corrections added in round 2 (each is also pinned against regenie's own output in tests/test_reference_pin.py) ----
def test_sparse_and_dense_branch_are_one_number_without_masked_samples():
    X, Y, blup, mask, res, scf, scale_Y, G = _problem(3, miss_y=False)
    G[1, ::7] = np.nan
    ref = s2.score_qt_block_ref(G, X, res, mask, scf)
    dense = s2.score_qt_block(G, X, res, mask, scf)
    assert 0 < ref["sparse"].sum() < G.shape[0]
    assert np.allclose(ref["stats"], dense["stats"], rtol=1e-10) and np.allclose(ref["bhat"], dense["bhat"], rtol=1e-10)
    # with masked samples the sparse branch is the reference's approximation: a different number, the dense one is mask^T r^2
    X, Y, blup, mask, res, scf, scale_Y, G = _problem(3, miss_y=True)
    ref = s2.score_qt_block_ref(G, X, res, mask, scf)
    dense = s2.score_qt_block(G, X, res, mask, scf)
    sp = ref["sparse"] == 1
    assert np.abs(ref["stats"][sp] / dense["stats"][sp] - 1).max() > 1e-4
    assert np.allclose(ref["stats"][~sp], dense["stats"][~sp], rtol=1e-12)


def _bt_problem(seed, n=600, C=3):
    from oracle import regenie_step1 as orc
    rng = np.random.default_rng(seed)
    X = np.linalg.qr(np.column_stack([np.ones(n), rng.normal(size=(n, C - 1))]))[0]
    off = 0.3 * rng.normal(size=n)
    g = rng.binomial(2, 0.2, size=n).astype(np.float64)
    eta = -1.0 + 12 * X[:, 1] + off + 0.5 * (g - g.mean())
    y = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(np.float64)
    mask = (rng.random(n) > 0.05)
    return orc, X, off, g, y, mask


def test_firth_fits_maximise_the_penalised_likelihood():
    """firth_snp / firth_fit against brute force: the reported maximisers beat every nearby point of the penalised log-likelihood, the
    1-parameter fit equals firth_fit on a single column, and the LRT is the drop in penalised deviance."""
    from scipy.optimize import minimize_scalar
    from oracle import regenie_step2_bt as bt
    orc, X, off, g, y, mask = _bt_problem(5)
    m = mask.astype(bool)
    gc = np.where(m, g - g[m].mean(), 0.0)

    def pen_dev(b):
        p = orc.get_pvec(off[m] + gc[m] * b)
        return -2 * np.sum(np.where(y[m] == 0, np.log(1 - p), np.log(p))) - np.log(np.sum(gc[m] ** 2 * p * (1 - p)))

    beta, se, lrt = bt.firth_snp(y, gc, mask.astype(float), off, root=True)
    # regenie's own answer (its first solver restated to the letter, stopped at |modified score| < 2.5e-4) lies within that tolerance of the maximiser
    b_reg, se_reg, lrt_reg = bt.firth_snp(y, gc, mask.astype(float), off)
    assert abs(b_reg - beta) < 2.5e-4 * se * se * 1.01 and abs(lrt_reg - lrt) < 1e-6 and se_reg == pytest.approx(se, rel=1e-4)
    best = minimize_scalar(pen_dev, bracket=(beta - 1, beta + 1), tol=1e-13)
    assert beta == pytest.approx(best.x, abs=1e-7) and lrt == pytest.approx(pen_dev(0.0) - pen_dev(beta), rel=1e-10)
    one = bt.firth_fit(y, gc[:, None], mask.astype(float), off, np.zeros(1), 1, maxstep=5.0)
    assert one[0][0] == pytest.approx(beta, abs=1e-8) and np.sqrt(one[2][0, 0]) == pytest.approx(se, rel=1e-8)
    # the covariate-only model: firth_null and firth_fit with every column free agree; constrained fit has the last coefficient untouched
    bn = bt.firth_null(y, X, mask, off, np.zeros(X.shape[1]), stop_tol=0.0)
    ff = bt.firth_fit(y, X, mask.astype(float), off, np.zeros(X.shape[1]), X.shape[1])
    assert np.allclose(bn, ff[0], atol=1e-8)
    assert np.abs(bt.firth_null(y, X, mask, off, np.zeros(X.shape[1])) - bn).max() < 2e-5 * np.abs(bn).max()       # regenie's stopping rule (50 numtol on the modified score)
    Xg = np.column_stack([X, g])
    nul = bt.firth_fit(y, Xg, mask.astype(float), off, np.concatenate([bn, [0.0]]), X.shape[1])
    full = bt.firth_fit(y, Xg, mask.astype(float), off, nul[0], X.shape[1] + 1, maxstep=5.0)
    assert nul[0][-1] == 0.0 and full[1] <= nul[1] + 1e-9


def test_spa_agrees_with_the_normal_tail_when_the_statistic_is_nearly_normal():
    """Balanced trait, common variant, |z| around 2: the saddlepoint p-value is within a few per cent of the normal one and symmetric in the
    sign of the statistic; the fast form (carriers exact, the rest normal) is close to the full one."""
    from scipy.stats import norm
    from oracle import regenie_step2_bt as bt
    orc, X, off, g, y, mask = _bt_problem(9, n=4000)
    opt = orc.Step1Options(bed="x", pheno_file="x", bt=True)
    null = bt.null_logistic(y, X, mask, off, opt)
    out = bt.score_bt(g, X, y, mask.astype(float), null)
    for z in (2.3, -2.3):
        sp = bt.spa_test(z, out["denum"], out["Gres"], null, mask.astype(float))
        fast = bt.spa_test(z, out["denum"], out["Gres"], null, mask.astype(float), carriers=np.flatnonzero(g != 0))
        pn = 2 * norm.sf(abs(z))
        assert 10 ** -sp["logp"] == pytest.approx(pn, rel=0.05) and 10 ** -fast["logp"] == pytest.approx(10 ** -sp["logp"], rel=0.02)
        assert np.sign(sp["bhat"]) == np.sign(z) and sp["se"] == pytest.approx(1 / np.sqrt(out["denum"]))
    a = bt.spa_test(2.3, out["denum"], out["Gres"], null, mask.astype(float))
    b = bt.spa_test(-2.3, out["denum"], out["Gres"], null, mask.astype(float))
    assert a["chisq"] == pytest.approx(b["chisq"], rel=1e-12)
