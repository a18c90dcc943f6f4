"""The null Firth model of `regenie-amd --step 2 --bt --firth` (host C++, regenie_amd/host/driver_models.cpp: firth_null_fit / firth_fit_cols),
compiled here with g++ into a small harness (no GPU involved) and held to oracle/regenie_step2_bt.py, which is pinned to regenie itself:
with samples masked for the trait the reference keeps their rows in X^T W X with weight 1 (get_wvec, Step1_Models.cpp:1809-1811; fit_firth_nr,
Step2_Models.cpp:1287-1290) -- penalty, hat diagonal and Newton matrix see them, likelihood and score do not (tests/golden/fuzz_log.md: regenie's
multi-trait run and its own single-trait run of the same trait differ by 1.2e-3 in an approximate-Firth BETA; the oracle reproduces the former)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import regenie_step2_bt as bt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = r'''
#include "driver.h"
extern "C" int firth_null(const double* y, const double* X, const uint8_t* mask, const double* offset, int64_t n, int C, double* beta) {
  std::vector<double> b(beta, beta + C);
  const bool ok = rgdrv::firth_null_fit(y, X, mask, offset, n, C, b);
  for (int c = 0; c < C; ++c) beta[c] = b[c];
  return ok ? 1 : 0;
}
extern "C" int firth_exact(const double* y, const double* X, const double* g, const uint8_t* mask, const double* offset, int64_t n, int C, double* beta, double* dev) {
  std::vector<const double*> cols(C + 1);
  for (int c = 0; c < C; ++c) cols[c] = X + (size_t)c * n;
  cols[C] = g;
  std::vector<double> b(beta, beta + C + 1);
  const bool ok = rgdrv::firth_fit_cols(y, cols, mask, offset, n, C + 1, 5.0, b, dev, nullptr);
  for (int c = 0; c <= C; ++c) beta[c] = b[c];
  return ok ? 1 : 0;
}
'''


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("firthnull")
    src = d / "h.cpp"
    src.write_text(HARNESS)
    so = d / "libfirthnull.so"
    host = os.path.join(ROOT, "regenie_amd", "host")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I" + host, os.path.join(host, "driver_models.cpp"), str(src), "-o", str(so)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return C.CDLL(str(so))


def _case(seed, n, c, miss):
    rng = np.random.default_rng(seed)
    Xr = np.column_stack([np.ones(n), rng.standard_normal((n, c - 1))])
    X = np.linalg.qr(Xr)[0]                                                      # orthonormal columns over ALL analysed samples, as getBasis leaves them
    lp = -1.5 + 0.8 * Xr[:, 1] - 0.5 * Xr[:, 2]
    y = (rng.random(n) < 1 / (1 + np.exp(-lp))).astype(np.float64)
    mask = rng.random(n) >= miss
    offset = 0.3 * rng.standard_normal(n)
    offset[~mask] = np.nan                                                       # NA predictions belong to the masked samples
    return X, y, mask, offset


@pytest.mark.parametrize("seed,n,c,miss", [(1, 400, 3, 0.0), (2, 300, 4, 0.06), (3, 900, 6, 0.15), (4, 150, 3, 0.3)])
def test_null_firth_fit_follows_the_oracle_with_masked_samples(lib, seed, n, c, miss):
    X, y, mask, offset = _case(seed, n, c, miss)
    start = np.zeros(c)
    want = bt.firth_null(y, X, mask, offset, start)
    assert want is not None
    Xf = np.asfortranarray(X)
    beta = start.copy()
    off = np.nan_to_num(offset)
    ok = lib.firth_null(y.ctypes.data_as(C.c_void_p), Xf.ctypes.data_as(C.c_void_p), mask.astype(np.uint8).ctypes.data_as(C.c_void_p),
                        off.ctypes.data_as(C.c_void_p), C.c_int64(n), C.c_int(c), beta.ctypes.data_as(C.c_void_p))
    assert ok == 1
    assert np.abs(beta - want).max() <= 1e-7 * np.abs(want).max()
    if miss > 0:      # and the masked rows matter: the fit on the unmasked samples alone is another point
        alone = bt.firth_null(y[mask], X[mask], np.ones(int(mask.sum()), bool), off[mask], start)
        assert np.abs(alone - want).max() > 1e-4 * np.abs(want).max()


def test_full_firth_fit_with_a_genotype_column_follows_the_oracle(lib):
    X, y, mask, offset = _case(9, 500, 4, 0.1)
    rng = np.random.default_rng(99)
    g = rng.binomial(2, 0.2, 500).astype(np.float64)
    off = np.nan_to_num(offset)
    want = bt.firth_fit(y, np.column_stack([X, g]), mask, off, np.zeros(5), 5, maxstep=5.0)
    assert want is not None
    beta, dev = np.zeros(5), np.zeros(1)
    Xf = np.asfortranarray(X)
    ok = lib.firth_exact(y.ctypes.data_as(C.c_void_p), Xf.ctypes.data_as(C.c_void_p), g.ctypes.data_as(C.c_void_p), mask.astype(np.uint8).ctypes.data_as(C.c_void_p),
                         off.ctypes.data_as(C.c_void_p), C.c_int64(500), C.c_int(4), beta.ctypes.data_as(C.c_void_p), dev.ctypes.data_as(C.c_void_p))
    assert ok == 1
    assert np.abs(beta - want[0]).max() <= 1e-7 * np.abs(want[0]).max() and abs(dev[0] - want[1]) <= 1e-8 * abs(want[1])
