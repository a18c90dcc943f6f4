"""The null Cox model of `regenie-amd --step 1 --t2e` (host C++, regenie_amd/host/driver_models.cpp): the coordinate descent the
reference tries first (cox_ridge at lambda = 0) and the Newton fall-back (cox_firth.cpp without the Firth term, fit_null_cox,
Step1_Models.cpp:415-436), compiled here with g++ into a small harness (no GPU involved) and checked against the oracle's gradient of
the log partial likelihood and against each other -- on correlated covariates, where the diagonal-Hessian descent is slow."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import regenie_step1_t2e as t2e

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = r'''
#include "driver.h"
extern "C" int cox_null_both(const double* time, const double* event, const uint8_t* mask, const double* X, int64_t N, int C, int niter,
                             double* eta_cd, double* eta_nr, int* ok) {
  rgdrv::Params p;
  p.niter_max = niter;
  std::vector<double> e1, e2;
  ok[0] = rgdrv::cox_null_fit(time, event, mask, X, N, C, p, e1);
  ok[1] = rgdrv::cox_null_newton(time, event, mask, X, N, C, p, 2.5e-4, e2);
  std::vector<double> e3;
  ok[2] = rgdrv::cox_null_newton(time, event, mask, X, N, C, p, 0.0, e3) ? 1 : 0;   // the second form of the fall-back runs too
  for (int64_t i = 0; i < N; ++i) { eta_cd[i] = e1[i]; eta_nr[i] = e2[i]; }
  return 0;
}
'''


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("coxnull")
    src = d / "h.cpp"
    src.write_text(HARNESS)
    so = d / "libcoxnull.so"
    host = os.path.join(ROOT, "regenie_amd", "host")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I" + host, os.path.join(host, "driver_models.cpp"), str(src), "-o", str(so)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return C.CDLL(str(so))


def _case(seed, n, rho):
    rng = np.random.default_rng(seed)
    z = rng.standard_normal((n, 1))
    Xr = np.sqrt(rho) * z + np.sqrt(1 - rho) * rng.standard_normal((n, 5))        # five covariates with pairwise correlation rho
    Xr = np.column_stack([Xr, (rng.random(n) < 0.5).astype(float)])
    mask = rng.random(n) > 0.05
    Xc = (Xr - Xr.mean(axis=0)) * mask[:, None]
    X = np.linalg.qr(Xc)[0] * np.sqrt(n)                                         # an orthogonal basis of the covariates, as getBasis leaves them
    lp = 0.5 * Xr[:, 0] - 0.3 * Xr[:, 1] + 0.4 * Xr[:, 5]
    t_ev = rng.exponential(1.0, n) * np.exp(-lp)
    t_c = rng.exponential(2.0, n)
    time = np.round(np.minimum(t_ev, t_c), 2) + 0.01                               # tied event times
    event = (t_ev <= t_c).astype(float)
    return np.asfortranarray(X), time, event, mask


def _score(X, eta, time, event, mask):
    sd = t2e.SurvivalData(time, event, mask, True)
    f = t2e.CoxRidge(sd, np.zeros((len(time), 0)), np.where(mask, eta, 0.0), mask, 0.0, 1, 1, 1e-6)
    f.cox_grad(sd)
    return np.abs(f.gradient @ X).max(), f.cox_deviance(sd)


@pytest.mark.parametrize("seed,rho", [(1, 0.0), (2, 0.9)])
def test_newton_fallback_finds_the_maximiser(lib, seed, rho):
    n = 3000
    X, time, event, mask = _case(seed, n, rho)
    m8 = mask.astype(np.uint8)
    eta_cd, eta_nr = np.zeros(n), np.zeros(n)
    ok = (C.c_int * 3)()
    lib.cox_null_both(time.ctypes.data_as(C.c_void_p), event.ctypes.data_as(C.c_void_p), m8.ctypes.data_as(C.c_void_p), X.ctypes.data_as(C.c_void_p),
                      C.c_int64(n), C.c_int(X.shape[1]), C.c_int(50), eta_cd.ctypes.data_as(C.c_void_p), eta_nr.ctypes.data_as(C.c_void_p), ok)
    assert ok[1] == 1 and ok[2] == 1
    s_nr, dev_nr = _score(X, eta_nr, time, event, mask)
    assert s_nr < 2.5e-4                                   # the stopping rule of cox_firth::fit, evaluated by the oracle's gradient
    assert (eta_nr[~mask] == 0).all()
    if ok[0]:                                              # the descent converged too: same model up to its looser stopping rule
        s_cd, dev_cd = _score(X, eta_cd, time, event, mask)
        assert dev_nr <= dev_cd + 1e-9                     # Newton ends at least as deep
        assert dev_cd - dev_nr < 1e-3 * abs(dev_nr)
        assert np.abs(eta_cd - eta_nr).max() < 0.05 * np.abs(eta_nr).max()


def test_descent_failure_is_reported_not_fatal(lib):
    """One pass is not enough for the coordinate descent: it reports failure (the driver then takes the
    Newton route, and only drops the trait if that fails as well)."""
    n = 2000
    X, time, event, mask = _case(3, n, 0.9)
    X = np.asfortranarray(np.column_stack([X[:, 0] + 0.2 * X[:, 1], X[:, 1], X[:, 2] + 0.5 * X[:, 0]]))     # correlated columns
    m8 = mask.astype(np.uint8)
    eta_cd, eta_nr = np.zeros(n), np.zeros(n)
    ok = (C.c_int * 3)()
    lib.cox_null_both(time.ctypes.data_as(C.c_void_p), event.ctypes.data_as(C.c_void_p), m8.ctypes.data_as(C.c_void_p), X.ctypes.data_as(C.c_void_p),
                      C.c_int64(n), C.c_int(3), C.c_int(1), eta_cd.ctypes.data_as(C.c_void_p), eta_nr.ctypes.data_as(C.c_void_p), ok)
    assert ok[0] == 0                                      # niter_max = 1: a single pass cannot meet the stopping rule
    # a single Newton step does not converge either at this tolerance; if it does, it ends below the null deviance
    _, dev0 = _score(X, np.zeros(n), time, event, mask)
    if ok[1]:
        assert _score(X, eta_nr, time, event, mask)[1] < dev0
