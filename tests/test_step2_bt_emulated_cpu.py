"""The device code of regenie_amd/csrc/step2_bt.hip EXECUTED IN THIS CONTAINER: the source is compiled for the host against tests/hipcpu (a stand-in
for the HIP runtime that runs a kernel one workgroup at a time, its work-items as fibers: __syncthreads, 64-lane __shfl_down and __shared__ arrays
behave as on the device), with plain-loop stand-ins for the i8 contraction entries of step2_qt.hip it calls (tests/hipcpu/s2_contract_host.cpp).
What runs is the library's own rg_s2_bt_set_null / rg_s2_bt_score_* / rg_s2_bt_correct with k_bt_prep, k_bt_firth1, k_bt_spa and k_bt_count_two as
written -- so the bodies of the GPU tests of tests/test_step2_bt_gpu.py (score test, approximate Firth, saddlepoint, full and carriers-only forms, and
the reference's minor-allele flip) can be held to the oracle without a GPU.  It does not replace those tests: the compiler, the matrix-core
contraction and the memory system of the device are not under test here, the arithmetic and the control flow of these kernels are."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emulated(tmp_path_factory):
    d = tmp_path_factory.mktemp("s2emu")
    so = str(d / "libs2emu.so")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-w", "-shared", "-x", "c++", "-I" + os.path.join(ROOT, "tests", "hipcpu"),
                        os.path.join(ROOT, "regenie_amd", "csrc", "step2_bt.hip"), os.path.join(ROOT, "tests", "hipcpu", "s2_contract_host.cpp"), "-o", so],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lib = C.CDLL(so)
    lib.rg_s2_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int32, C.c_int32]
    lib.rg_s2_destroy.argtypes = [C.c_void_p]
    lib.rg_s2_destroy.restype = None
    lib.rg_s2_last_error.argtypes = [C.c_void_p]
    lib.rg_s2_last_error.restype = C.c_char_p
    lib.rg_s2_set_sparse_rule.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_int32]
    lib.rg_s2_bt_set_null.argtypes = [C.c_void_p, C.c_void_p]
    lib.rg_s2_bt_score_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_void_p]
    lib.rg_s2_bt_score_int.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_void_p]
    lib.rg_s2_bt_correct.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.rg_s2_last_kernel_ms.argtypes = [C.c_void_p]
    lib.rg_s2_last_kernel_ms.restype = C.c_double
    return lib


@pytest.fixture()
def gpu_tests(emulated, monkeypatch):
    import regenie_amd.step2 as step2
    monkeypatch.setattr(step2, "load_library", lambda: emulated)
    from tests import test_step2_bt_gpu as t
    return t


@pytest.mark.parametrize("route", ["packed", "int"])
def test_score_and_corrections_on_the_emulated_device(gpu_tests, route):
    gpu_tests.test_bt_score_and_corrections_against_the_oracle(route)          # the GPU test's own body, at its own size (3,001 samples, 96 pairs x 2 forms x 2 corrections)


@pytest.mark.parametrize("route", ["packed", "int"])
def test_minor_allele_flip_on_the_emulated_device(gpu_tests, route):
    gpu_tests.test_bt_corrections_on_the_allele_the_reference_tests(route)


def test_usage_errors_on_the_emulated_device(gpu_tests):
    gpu_tests.test_bt_usage_errors()


def _one_shape(t, seed, n, Cc, P, bs, route, miss_y, nrule):
    from regenie_amd.step2 import BT_FIRTH_APPROX, BT_SPA, Step2QT
    from oracle import regenie_step2_bt as bt
    rng = np.random.default_rng(seed)
    X = np.linalg.qr(np.column_stack([np.ones(n)] + ([rng.normal(size=(n, Cc - 1))] if Cc > 1 else [])))[0]
    off = 0.3 * rng.normal(size=(n, P))
    maf = rng.uniform(0.03, 0.5, size=bs)
    G = rng.binomial(2, maf[:, None], size=(bs, n)).astype(np.float64)
    G[::3] = 2.0 - G[::3]
    lin = (X[:, [1]] * 3 if Cc > 1 else 0) + off
    y = (rng.random((n, P)) < 1 / (1 + np.exp(-(-0.5 + lin)))).astype(np.float64)
    mask = rng.random((n, P)) > miss_y
    G[rng.random(G.shape) < 0.02] = np.nan
    if route == "int":
        pick = (rng.random(G.shape) < 0.05) & ~np.isnan(G)
        v = np.clip(np.nan_to_num(G) * 255 + rng.integers(-60, 61, G.shape), 0, 510)
        G = np.where(pick, v / 255.0, G)
    nulls, fo = t._nulls(X, off, y, mask)
    fitted = np.array([nl["p"] for nl in nulls])
    with Step2QT(n, X.shape[1], P) as s2:
        s2.set_sparse_rule(nrule, 0.5, False)
        s2.bt_set_null(X.T, y.T, mask.T, fitted, firth_offset=fo)
        got = s2.bt_score_packed(t._pack(G)) if route == "packed" else s2.bt_score_int(np.where(np.isnan(G), 0xFFFF, np.rint(np.nan_to_num(G) * 255)).astype(np.uint16), 255)
        pairs, gi, sg = [], [], []
        for j in range(bs):
            gk, flipped = bt.flip_geno(np.where(np.isnan(G[j]), -3.0, G[j]))
            obs = gk >= 0
            g = np.where(obs, gk, gk[obs].mean())
            sgn = -1.0 if flipped else 1.0
            gi.append(g); sg.append(sgn)
            sparse = int((g != 0).sum()) <= 0.5 * nrule
            assert bool(got["sparse"][j]) == sparse, ("sparse", j, flipped)
            for q in range(P):
                ref = bt.score_bt(g, X, y[:, q], mask[:, q].astype(float), nulls[q], sparse=sparse)
                if ref is None:
                    assert got["test_ignored"][j, q]; continue
                assert not got["test_ignored"][j, q]
                assert got["stats"][j, q] == pytest.approx(sgn * ref["stats"], rel=1e-8, abs=1e-9)
                assert got["denum"][j, q] == pytest.approx(ref["denum"], rel=1e-8)
                pairs.append((j, q, ref))
        var = np.array([p[0] for p in pairs], np.int32); tr = np.array([p[1] for p in pairs], np.int32)
        fast = np.array([int(got["sparse"][p[0]]) for p in pairs], np.uint8)
        fc = s2.bt_correct(BT_FIRTH_APPROX, var, tr, fast)
        sc = s2.bt_correct(BT_SPA, var, tr, fast)
    nf = ns = 0
    for tt, (j, q, ref) in enumerate(pairs):
        g, m, sgn = gi[j], mask[:, q].astype(float), sg[j]
        is_fast = bool(fast[tt])
        want = bt.approx_firth(g, X, y[:, q], m, nulls[q], fo[q], sparse=is_fast, mac=0 if is_fast else None)
        if want is None:
            assert fc["fail"][tt] == 1, ("firth fail", j, q)
        else:
            assert fc["fail"][tt] == 0, ("firth ok", j, q)
            assert fc["beta"][tt] == pytest.approx(sgn * want["bhat"], rel=1e-5, abs=1e-7), ("firth beta", j, q, is_fast)
            assert fc["chisq"][tt] == pytest.approx(want["chisq"], rel=1e-5, abs=1e-8)
            nf += 1
        if abs(ref["stats"]) < 0.1: continue
        wsp = bt.spa_test(ref["stats"], ref["denum"], ref["Gres"], nulls[q], m, carriers=np.flatnonzero(g != 0) if is_fast else None)
        if wsp is None:
            assert sc["fail"][tt] == 1, ("spa fail", j, q)
        else:
            assert sc["fail"][tt] == 0, ("spa ok", j, q)
            assert sc["logp"][tt] == pytest.approx(wsp["logp"], rel=1e-5, abs=1e-8), ("spa logp", j, q, is_fast, ref["stats"])
            ns += 1
    return nf, ns



@pytest.mark.parametrize("n", [40, 64, 65, 255, 257, 700])
def test_shapes_the_gpu_tests_do_not_visit(gpu_tests, n):
    """Sample counts around the wavefront and workgroup sizes, 1 - 5 covariates, one and three traits, masks, both input routes, every third variant
    counting its major allele, the rule's n_samples above the analysed samples: score test, sparse verdict and both corrections of every pair
    against the oracle (192 such combinations were run when this was written; a sixth of them stay here)."""
    k = 0
    for Cc, P, route, miss_y in ((1, 1, "packed", 0.0), (2, 3, "int", 0.1), (5, 3, "packed", 0.1), (5, 1, "int", 0.0)):
        k += 1
        nf, ns = _one_shape(gpu_tests, 2000 + 10 * n + k, n, Cc, P, 9, route, miss_y, n + (k % 3) * 50)
        assert nf >= 7 * P and ns >= 5 * P


REGENIE = os.path.join(ROOT, "oracle", "_ref", "regenie")


@pytest.mark.skipif(not os.path.exists(REGENIE), reason="oracle/_ref/regenie not built (make -C oracle)")
def test_the_driver_on_emulated_kernels_beside_regenie_itself(tmp_path, monkeypatch):
    """`regenie-amd --step 2 --bt --firth --approx | --spa` -- the driver as shipped, linked with step2_bt.hip on the host stand-in (tests/hipcpu/emubuild.py) --
    beside regenie on two drawn cases of tests/golden/fuzz_oracle_vs_reference.py, from the .bed, from the same genotypes as BGEN dosages and (third case) as a .pgen with a dosage track; the second case
    runs with --ref-first, so every variant counts its major allele and the carriers of the fast forms are those of 2 - g (flip_geno).  The 100-case run of
    the round is tests/golden/fuzz_driver_log.md."""
    from tests.golden import fuzz_oracle_vs_reference as fz
    from tests.hipcpu.emubuild import build_bt_step2_driver
    monkeypatch.setattr(fz, "BIN", build_bt_step2_driver(str(tmp_path / "build")))
    for k, v in dict(FUZZ_ROUTES="bt_loocv", FUZZ_BT_STEP2="3", FUZZ_DRIVER="1", FUZZ_DRIVER_BT_ONLY="1", RG_S2_BGEN_ROWS="1").items():
        monkeypatch.setenv(k, v)
    for seed in (3, 6):
        line, ok = fz.run_one(seed, str(tmp_path))
        assert ok, line
        assert "from BGEN dosages" in line
    monkeypatch.setenv("FUZZ_BT_STEP2", "4")             # and from a .pgen with a dosage track: the reader's zero count decides `sparse` (before the flip)
    line, ok = fz.run_one(5, str(tmp_path))
    assert ok and "from a .pgen with dosages" in line, line
    # ... and not only within regenie's stopping tolerances: with the sparse form's numerator, regenie's fit_firth_pseudo in the Firth kernel and its stopping rule
    # in the null Firth model, the driver's files of this case -- uncorrected, saddlepoint and approximate-Firth rows, from the .bed and from the .pgen --
    # are regenie's files byte for byte (two traits, 5 % of the phenotypes missing, 2 % of the genotypes)
    d = os.path.join(str(tmp_path), "c5")
    pairs = [(f, "s" + f[1:]) for f in sorted(os.listdir(d)) if f.startswith("d2") and f.endswith(".regenie")]
    assert len(pairs) == 10, pairs                      # two traits x (score test, approximate Firth, saddlepoint from the .bed; the two corrections from the .pgen)
    for mine, theirs in pairs:
        assert open(os.path.join(d, mine)).read() == open(os.path.join(d, theirs)).read(), (mine, theirs)
