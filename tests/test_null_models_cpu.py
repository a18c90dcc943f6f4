"""The null logistic / Poisson models of `regenie-amd` (host C++, regenie_amd/host/driver_models.cpp: fit_logistic, fit_poisson -- Step 1's
covariate-only fits, row a14, and Step 2's fits with the LOCO prediction as offset), compiled with g++ into a small harness (no GPU) and held
to the oracle (oracle/regenie_step2_bt.py: null_logistic / null_poisson, pinned to regenie's own runs) on drawn data with samples masked for
the trait and NA offsets at those samples."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import regenie_step1 as orc
from oracle import regenie_step2_bt as bt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = r'''
#include "driver.h"
extern "C" int null_fit(int family, const double* y, const double* X, const uint8_t* mask, const double* offset, int64_t N, int C, double* pv, int niter) {
  rgdrv::Params p;
  if (niter > 0) p.niter_max = niter;
  std::vector<double> eta, fitted;
  rgdrv::LogisticState st;      // as the driver calls it: the second attempt resumes from the first one's state
  const bool ok = family == 0 ? rgdrv::fit_logistic(y, X, mask, N, C, p, true, eta, offset, &fitted, nullptr, &st) || rgdrv::fit_logistic(y, X, mask, N, C, p, false, eta, offset, &fitted, nullptr, &st)
                              : rgdrv::fit_poisson(y, X, mask, N, C, p, eta, offset, &fitted);
  if (ok) for (int64_t i = 0; i < N; ++i) pv[i] = fitted[i];
  return ok ? 1 : 0;
}
'''


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("nullmodels")
    src = d / "h.cpp"
    src.write_text(HARNESS)
    so = d / "libnull.so"
    host = os.path.join(ROOT, "regenie_amd", "host")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I" + host, os.path.join(host, "driver_models.cpp"), str(src), "-o", str(so)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return C.CDLL(str(so))


@pytest.mark.parametrize("family,seed,n,c,miss", [(0, 1, 500, 3, 0.0), (0, 2, 350, 5, 0.08), (0, 3, 1200, 4, 0.2), (1, 4, 600, 3, 0.0), (1, 5, 400, 4, 0.1)])
def test_null_model_fits_follow_the_oracle(lib, family, seed, n, c, miss):
    rng = np.random.default_rng(seed)
    Xr = np.column_stack([np.ones(n), rng.standard_normal((n, c - 1))])
    X = np.linalg.qr(Xr)[0]
    lp = 0.7 * Xr[:, 1] - 0.4 * Xr[:, 2]
    y = (rng.random(n) < 1 / (1 + np.exp(-(lp - 1.0)))).astype(np.float64) if family == 0 else rng.poisson(np.exp(0.2 + 0.3 * lp)).astype(np.float64)
    mask = rng.random(n) >= miss
    offset = 0.25 * rng.standard_normal(n)
    offset[~mask] = np.nan
    opt = orc.Step1Options()
    want = bt.null_logistic(y, X, mask, offset, opt) if family == 0 else bt.null_poisson(y, X, mask, offset, opt)
    assert want is not None
    Xf = np.asfortranarray(X)
    off = np.nan_to_num(offset)
    pv = np.zeros(n)
    ok = lib.null_fit(C.c_int(family), y.ctypes.data_as(C.c_void_p), Xf.ctypes.data_as(C.c_void_p), mask.astype(np.uint8).ctypes.data_as(C.c_void_p),
                      off.ctypes.data_as(C.c_void_p), C.c_int64(n), C.c_int(c), pv.ctypes.data_as(C.c_void_p), C.c_int(0))
    assert ok == 1
    assert np.abs(pv[mask] - want["p"][mask]).max() <= 1e-9 * max(1.0, np.abs(want["p"][mask]).max())


def test_second_attempt_of_the_null_logistic_fit_resumes_from_the_first(lib):
    """`fit_logistic(.., true, ..) || fit_logistic(.., false, ..)` on the reference's in-place arguments (Step1_Models.cpp:88): with --niter k the model gets 2 k
    Newton steps.  Data on which k = 2 does not converge from a fresh start but 2 + 2 does: the driver's chained call and the oracle's agree, and both differ
    from a restarted second attempt (which fails)."""
    rng = np.random.default_rng(11)
    n, c = 600, 4
    Xr = np.column_stack([np.ones(n), rng.standard_normal((n, c - 1))])
    X = np.linalg.qr(Xr)[0]
    y = (rng.random(n) < 1 / (1 + np.exp(-(1.2 * Xr[:, 1] - 0.9 * Xr[:, 2] - 0.8)))).astype(np.float64)
    mask = np.ones(n, bool)
    for k in range(2, 8):                                                          # the smallest --niter whose 2 k chained steps converge
        opt = orc.Step1Options(niter_max=k)
        want = bt.null_logistic(y, X, mask, np.zeros(n), opt)
        if want is not None:
            break
    assert want is not None
    eta0 = np.zeros(n)
    ok1, b1, p1, e1 = orc.fit_logistic(y, X, np.zeros(n), mask, orc.get_pvec(eta0), eta0, np.zeros(c), opt, True, 1e-6)
    assert not ok1                                                                 # k steps from a fresh start are not enough ...
    ok2, *_ = orc.fit_logistic(y, X, np.zeros(n), mask, orc.get_pvec(eta0), eta0.copy(), np.zeros(c), opt, False, 1e-6)
    assert not ok2                                                                 # ... nor are k more from another fresh start
    pv = np.zeros(n)
    Xf = np.asfortranarray(X)
    off = np.zeros(n)
    ok = lib.null_fit(C.c_int(0), y.ctypes.data_as(C.c_void_p), Xf.ctypes.data_as(C.c_void_p), mask.astype(np.uint8).ctypes.data_as(C.c_void_p),
                      off.ctypes.data_as(C.c_void_p), C.c_int64(n), C.c_int(c), pv.ctypes.data_as(C.c_void_p), C.c_int(k))
    assert ok == 1 and np.abs(pv - want["p"]).max() <= 1e-9
