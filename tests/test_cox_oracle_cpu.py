"""Closed-form checks of the time-to-event oracle (oracle/regenie_step1_t2e.py), independent of regenie's output files (CPU):
the log partial likelihood against a naive Breslow sum (with tied event times), its gradient and diagonal Hessian against finite
differences, and the coordinate descent's fixed point against a quasi-Newton minimiser of the same penalised objective."""
import numpy as np
import pytest

from oracle import regenie_step1_t2e as t2e


def _data(seed, n=300, p=4, ties=True, masked=0.1):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(n, p))
    t_ev = rng.exponential(1.0, n) * np.exp(-0.5 * X[:, 0])
    t_c = rng.exponential(1.5, n)
    time = np.minimum(t_ev, t_c)
    if ties:
        time = np.round(time, 1) + 0.1
    event = (t_ev <= t_c).astype(float)
    mask = rng.random(n) > masked
    return X, time, event, mask


def _breslow_loglik(eta, time, event, mask):
    """sum over events of eta_i - log sum_{j in risk set of t_i} exp(eta_j): Breslow's form, every tied event with the full risk set."""
    idx = np.flatnonzero(mask)
    ll = 0.0
    for i in idx:
        if event[i] == 1:
            risk = idx[time[idx] >= time[i]]
            ll += eta[i] - np.log(np.exp(eta[risk]).sum())
    return ll


@pytest.mark.parametrize("ties", [False, True])
def test_loglik_is_breslow(ties):
    X, time, event, mask = _data(3, ties=ties)
    sd = t2e.SurvivalData(time, event, mask, True)
    beta = np.array([0.3, -0.2, 0.1, 0.05])
    f = t2e.CoxRidge(sd, X, np.zeros(len(time)), mask, 0.0, 10, 10, 1e-6, beta_init=beta)
    # the oracle weighs every sample by 1 / neff and, in the risk-set sums, too: log(sum w e^eta) = log(sum e^eta) - log(neff) per event
    nev = int((event[mask] == 1).sum())
    want = (_breslow_loglik(np.where(mask, X @ beta, 0.0), time, event, mask) + nev * np.log(sd.neff)) / sd.neff
    assert f.cox_loglik(sd) == pytest.approx(want, rel=1e-12)


def test_gradient_and_diagonal_hessian_by_finite_differences():
    X, time, event, mask = _data(5, n=120, ties=True)
    n = len(time)
    sd = t2e.SurvivalData(time, event, mask, True)
    eta0 = np.where(mask, 0.4 * X[:, 0] - 0.1 * X[:, 1], 0.0)

    def ll(eta):
        f = t2e.CoxRidge(sd, np.zeros((n, 0)), eta, mask, 0.0, 1, 1, 1e-6)     # no columns: eta = the offset
        return f.cox_loglik(sd)

    f = t2e.CoxRidge(sd, np.zeros((n, 0)), eta0, mask, 0.0, 1, 1, 1e-6)
    f.cox_grad(sd)
    h = 1e-5
    for i in np.flatnonzero(mask)[:25]:
        e = np.zeros(n); e[i] = h
        g = (ll(eta0 + e) - ll(eta0 - e)) / (2 * h)
        d2 = (ll(eta0 + e) - 2 * ll(eta0) + ll(eta0 - e)) / h ** 2
        assert f.gradient[i] == pytest.approx(g, rel=1e-5, abs=1e-9)
        assert f.diag_hessian[i] == pytest.approx(d2, rel=2e-3, abs=1e-7)
    assert np.all(f.gradient[~mask] == 0) and np.all(f.diag_hessian[~mask] == 0)
    assert abs(f.gradient.sum()) < 1e-12          # the score of a shift of eta is zero


def test_coordinate_descent_reaches_the_penalised_optimum():
    from scipy.optimize import minimize
    X, time, event, mask = _data(11, n=400, p=3, ties=True)
    n = len(time)
    sd = t2e.SurvivalData(time, event, mask, True)
    lam = 0.05
    f = t2e.CoxRidge(sd, X, np.zeros(n), mask, lam, 500, 50, 1e-12)
    f.fit(sd, X, np.zeros(n), mask)

    def obj(b):
        return t2e.CoxRidge(sd, X, np.zeros(n), mask, lam, 1, 1, 1e-6, beta_init=b).objective[0]

    best = minimize(obj, np.zeros(3), method="BFGS", options={"gtol": 1e-10})
    # deviance + lam |beta|^2 / 2 is the objective the reference reports, but its coordinate update solves the score equation
    # X^T g - lam beta = 0 of  loglik - lam |beta|^2 / 2, i.e. of  deviance / 2 + lam |beta|^2 / 2: compare on that
    def obj2(b):
        c = t2e.CoxRidge(sd, X, np.zeros(n), mask, lam, 1, 1, 1e-6, beta_init=b)
        return c.deviance[0] / 2 + lam * (b ** 2).sum() / 2

    best2 = minimize(obj2, np.zeros(3), method="BFGS", options={"gtol": 1e-10})
    assert f.beta == pytest.approx(best2.x, rel=1e-4, abs=1e-6)
    assert best.fun <= obj(f.beta) + 1e-9         # and the reported objective is NOT what the iteration minimises (it is smaller elsewhere)
