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
