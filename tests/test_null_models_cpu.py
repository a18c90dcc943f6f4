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
extern "C" int null_fit(int family, const double* y, const double* X, const uint8_t* mask, const double* offset, int64_t N, int C, double* pv) {
  rgdrv::Params p;
  std::vector<double> eta, fitted;
  const bool ok = family == 0 ? rgdrv::fit_logistic(y, X, mask, N, C, p, true, eta, offset, &fitted) || rgdrv::fit_logistic(y, X, mask, N, C, p, false, eta, offset, &fitted)
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
                      off.ctypes.data_as(C.c_void_p), C.c_int64(n), C.c_int(c), pv.ctypes.data_as(C.c_void_p))
    assert ok == 1
    assert np.abs(pv[mask] - want["p"][mask]).max() <= 1e-9 * max(1.0, np.abs(want["p"][mask]).max())
