"""The panel-of-128 batched Cholesky of the level-0 ridge systems (regenie_amd/csrc/chol_p128.h: the kernel behind `rg_l0_blocks` for every block
whose right-hand sides are embedded) EXECUTED IN THIS CONTAINER: the header is compiled by g++ against tests/hipcpu (workgroups one at a time,
work-items as fibers; the fp64 matrix instruction, v_readlane and the global -> LDS copy restated from their lane layouts in
tests/hipcpu/chol128_host.cpp) and every factor, every forward-substituted right-hand side and every tile inverse is held to numpy's.
What this holds: the indexing of the kernel -- work placement, the ring of K / source / Linv units and its issue cursor, the transposed
accumulation and the triangular multiply, the paired row blocks of the diagonal blocks, the 2 x 2 tile factorization with its inverse, the
masks of embedded right-hand-side rows and of the identity padding, systems smaller than the batch's order (the reference's ragged last
block of a chromosome, src/Data.cpp:313-342 -> ridge_level_0, src/Step1_Models.cpp:484-505).  What it does not: waits, fences and caches of
the device -- tests/test_kernels_gpu.py and tests/test_step1_gpu.py run the same kernel on the MI355X."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("c128emu") / "libchol128_host.so")
    hc = os.path.join(ROOT, "tests", "hipcpu")
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-w", "-x", "c++", "-I" + hc, os.path.join(hc, "chol128_host.cpp"), "-o", so],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return ctypes.CDLL(so)


def _run(lib, n64, orders, R, embed, seed, skip_pad=1, with_F=False):
    rng = np.random.default_rng(seed)
    nouter = len(orders)
    S = np.full((nouter, n64, n64), 1e30)                 # whatever lies outside a system's own rows / columns must never be used
    F = np.full((nouter, n64, n64), 3e29) if with_F else None
    A_, B_ = [], []
    for o, n in enumerate(orders):
        G = rng.standard_normal((n, 3 * n))
        A = G @ G.T / n
        b = rng.standard_normal((embed, n))
        if with_F:                                        # X = S - F: the form of the callers that pass the fold matrix separately
            G2 = rng.standard_normal((n, n))
            A2, b2 = G2 @ G2.T / (4 * n), rng.standard_normal((embed, n))
            S[o, :n, :n], F[o, :n, :n] = A + A2, A2
            S[o, n:n + embed, :n], F[o, n:n + embed, :n] = b + b2, b2
        else:
            S[o, :n, :n] = A
            S[o, n:n + embed, :n] = b
        A_.append(A)
        B_.append(b)
    shift = np.array([0.5, 2.0, 7.0][:R])
    d_n = np.array(orders, dtype=np.int32)
    batch = nouter * R
    mats = np.full((batch, n64, n64), np.nan)
    dinv = np.full((batch, n64 // 64, 4096), np.nan)
    linv = np.full((batch, n64 // 128, 16384), np.nan)
    info = np.zeros(4, dtype=np.int32)
    P = ctypes.c_void_p
    nl = lib.c128_host_factor(P(S.ctypes.data), nouter, P(shift.ctypes.data), R, P(d_n.ctypes.data), n64, embed, skip_pad,
                              P(F.ctypes.data) if with_F else None, P(mats.ctypes.data), P(dinv.ctypes.data), P(linv.ctypes.data), P(info.ctypes.data))
    assert nl == n64 // 128 and info[0] == 0
    worst = 0.0
    for o, n in enumerate(orders):
        for r in range(R):
            k = o * R + r
            L = np.linalg.cholesky(A_[o] + shift[r] * np.eye(n))
            Y = np.linalg.solve(L, B_[o].T).T
            worst = max(worst, np.abs(np.tril(mats[k, :n, :n]) - L).max() / np.abs(L).max())
            if embed:
                worst = max(worst, np.abs(mats[k, n:n + embed, :n] - Y).max() / np.abs(Y).max())
            for t in range((n + 63) // 64):
                lo, hi = 64 * t, min(64 * t + 64, n)
                Ik = np.linalg.inv(L[lo:hi, lo:hi])
                worst = max(worst, np.abs(dinv[k, t].reshape(64, 64)[:hi - lo, :hi - lo] - Ik).max() / np.abs(Ik).max())
            for pnl in range((n + 127) // 128):           # the inverses of the diagonal 128-blocks: what the triangular multiply of a panel reads
                lo, hi = 128 * pnl, min(128 * pnl + 128, n)
                Ip = np.linalg.inv(L[lo:hi, lo:hi])
                got = np.tril(linv[k, pnl].reshape(128, 128)[:hi - lo, :hi - lo])
                worst = max(worst, np.abs(got - Ip).max() / np.abs(Ip).max())
    return worst


@pytest.mark.parametrize("n64,orders,R,embed,skip_pad,with_F", [
    (256, [200, 100], 2, 2, 1, False),             # two panels; a system of one panel beside one of two
    (512, [500, 300, 120], 2, 3, 1, True),         # four panels (K loops of 1 - 3 panels), X = S - F, systems of 4 / 3 / 1 panels
    (384, [382, 129], 1, 2, 0, False),             # right-hand-side rows in the last two rows of the order; no per-system skipping of the padding
    (128, [127, 64, 1], 3, 1, 1, False),           # a single panel: the diagonal block alone
])
def test_panel128_cholesky_on_the_host_stand_in(lib, n64, orders, R, embed, skip_pad, with_F):
    assert _run(lib, n64, orders, R, embed, seed=n64 + R, skip_pad=skip_pad, with_F=with_F) < 1e-12
