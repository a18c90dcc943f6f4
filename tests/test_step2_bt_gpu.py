"""Binary-trait Step 2 behind the C ABI (include/rg_step2.h: rg_s2_bt_set_null / rg_s2_bt_score_packed / rg_s2_bt_score_int /
rg_s2_bt_correct, regenie_amd/csrc/step2_bt.hip) against the oracle (oracle/regenie_step2_bt.py, pinned against regenie's own Step-2
output by tests/test_reference_pin.py): the score test of every (variant, trait), then the approximate Firth correction and the
saddlepoint approximation of EVERY testable pair -- each in its full form and in the reference's carriers-only form -- computed on the
device, one workgroup per pair."""
import numpy as np
import pytest

from oracle import regenie_step1 as orc
from oracle import regenie_step2_bt as bt

pytestmark = pytest.mark.gpu


def _problem(seed, n=3001, C=3, P=2, bs=48):
    rng = np.random.default_rng(seed)
    X = np.linalg.qr(np.column_stack([np.ones(n), rng.normal(size=(n, C - 1))]))[0]
    off = 0.3 * rng.normal(size=(n, P))
    maf = np.concatenate([rng.uniform(0.1, 0.4, size=bs // 2), rng.uniform(0.002, 0.01, size=bs - bs // 2)])     # common and rare (sparse) variants
    G = rng.binomial(2, maf[:, None], size=(bs, n)).astype(np.float64)
    eta = -1.2 + 10 * X[:, [1]] + off + 0.4 * (G[0] - G[0].mean())[:, None]
    y = (rng.random((n, P)) < 1 / (1 + np.exp(-eta))).astype(np.float64)
    mask = rng.random((n, P)) > 0.03
    G[rng.random(G.shape) < 0.01] = np.nan
    return X, off, G, y, mask


def _pack(G):
    bs, n = G.shape
    code = np.where(np.isnan(G), 1, np.where(G == 2, 0, np.where(G == 1, 2, 3))).astype(np.uint8)
    code = np.concatenate([code, np.zeros((bs, (-n) % 4), np.uint8)], axis=1).reshape(bs, -1, 4)
    return (code[:, :, 0] | (code[:, :, 1] << 2) | (code[:, :, 2] << 4) | (code[:, :, 3] << 6)).astype(np.uint8)


def _nulls(X, off, y, mask):
    opt = orc.Step1Options(bed="x", pheno_file="x", bt=True)
    nulls, fo = [], []
    for q in range(y.shape[1]):
        nl = bt.null_logistic(y[:, q], X, mask[:, q], off[:, q], opt)
        assert nl is not None
        bn = bt.firth_null(y[:, q], X, mask[:, q], np.where(mask[:, q], off[:, q], 0.0), nl["beta"])
        assert bn is not None
        nulls.append(nl)
        fo.append(np.where(mask[:, q], off[:, q], 0.0) + X @ bn)
    return nulls, np.array(fo)


@pytest.mark.parametrize("route", ["packed", "int"])
def test_bt_score_and_corrections_against_the_oracle(route):
    from regenie_amd.step2 import BT_FIRTH_APPROX, BT_SPA, Step2QT
    X, off, G, y, mask = _problem(11)
    bs, n = G.shape
    P = y.shape[1]
    nulls, fo = _nulls(X, off, y, mask)
    fitted = np.array([nl["p"] for nl in nulls])
    with Step2QT(n, X.shape[1], P) as s2:
        s2.set_sparse_rule(n, 0.5, False)
        s2.bt_set_null(X.T, y.T, mask.T, fitted, firth_offset=fo)
        if route == "packed":
            got = s2.bt_score_packed(_pack(G))
        else:
            G16 = np.where(np.isnan(G), 0xFFFF, np.nan_to_num(G) * 255).astype(np.uint16)
            got = s2.bt_score_int(G16, 255)
        gi = []          # mean-imputed genotypes
        pairs = []
        for j in range(bs):
            g = G[j].copy()
            mu = np.nanmean(g)
            g[np.isnan(g)] = mu
            gi.append(g)
            assert got["mean"][j] == pytest.approx(mu, rel=1e-12)
            nnz = int((g != 0).sum())
            assert bool(got["sparse"][j]) == (nnz <= 0.5 * n)
            for q in range(P):
                ref = bt.score_bt(g, X, y[:, q], mask[:, q].astype(float), nulls[q], sparse=nnz <= 0.5 * n)      # (the sparse form's numerator: GW . yres)
                assert ref is not None and not got["test_ignored"][j, q]
                assert got["stats"][j, q] == pytest.approx(ref["stats"], rel=1e-9, abs=1e-10)
                assert got["bhat"][j, q] == pytest.approx(ref["bhat"], rel=1e-9, abs=1e-12)
                assert got["denum"][j, q] == pytest.approx(ref["denum"], rel=1e-9)
                pairs.append((j, q, ref))
        # every pair, full form and carriers-only form (the latter only where the variant is sparse, as the reference)
        var = np.array([p[0] for p in pairs] * 2, np.int32)
        tr = np.array([p[1] for p in pairs] * 2, np.int32)
        fast = np.array([0] * len(pairs) + [int(got["sparse"][p[0]]) for p in pairs], np.uint8)
        fc = s2.bt_correct(BT_FIRTH_APPROX, var, tr, fast)
        sc = s2.bt_correct(BT_SPA, var, tr, fast)
    nfirth = nspa = 0
    for t, (j, q, ref) in enumerate(pairs + pairs):
        g, m = gi[j], mask[:, q].astype(float)
        is_fast = bool(fast[t])
        want = bt.approx_firth(g, X, y[:, q], m, nulls[q], fo[q], sparse=is_fast, mac=0 if is_fast else None)
        if want is None:
            assert fc["fail"][t] == 1
        else:
            assert fc["fail"][t] == 0
            assert fc["beta"][t] == pytest.approx(want["bhat"], rel=1e-6, abs=1e-8)
            assert fc["se"][t] == pytest.approx(want["se"], rel=1e-6)
            assert fc["chisq"][t] == pytest.approx(want["chisq"], rel=1e-6, abs=1e-9)
            nfirth += 1
        if abs(ref["stats"]) < 0.1:          # (a saddlepoint at t ~ 0, p ~ 1: -log10 p is all cancellation there -- 2e-6 apart at z = 0.006 when this source runs on
            continue                         # the host, tests/test_step2_bt_emulated_cpu.py -- and regenie corrects |z| above its threshold only)
        carriers = np.flatnonzero(g != 0) if is_fast else None
        wsp = bt.spa_test(ref["stats"], ref["denum"], ref["Gres"], nulls[q], m, carriers=carriers)
        if wsp is None:
            assert sc["fail"][t] == 1
        else:
            assert sc["fail"][t] == 0
            assert sc["logp"][t] == pytest.approx(wsp["logp"], rel=1e-6, abs=1e-9)
            assert sc["chisq"][t] == pytest.approx(wsp["chisq"], rel=1e-6, abs=1e-9)
            assert sc["beta"][t] == pytest.approx(wsp["bhat"], rel=1e-6, abs=1e-10) and sc["se"][t] == pytest.approx(wsp["se"], rel=1e-9)
            nspa += 1
    assert nfirth > len(pairs) and nspa > len(pairs)


@pytest.mark.parametrize("route", ["packed", "int"])
def test_bt_corrections_on_the_allele_the_reference_tests(route):
    """regenie tests the MINOR allele of an additive binary-trait test: a variant whose mean dosage exceeds 1 is flipped to 2 - g before the test
    and its BETA negated back (flip_geno, Geno.cpp:3150-3162; oracle/regenie_step2_bt.py: flip_geno, held to regenie's corrected rows on drawn cases,
    tests/golden/fuzz_log.md).  The printed score test does not see it; check_sparse_G's verdict and the carriers of the fast forms do.  Every other
    variant of this block counts its MAJOR allele: the library reports the statistics of the coding it was given, `sparse` and the carriers-only
    corrections of the coding the reference tests."""
    from regenie_amd.step2 import BT_FIRTH_APPROX, BT_SPA, Step2QT
    X, off, G, y, mask = _problem(23, bs=40)
    G[::2] = 2.0 - G[::2]                                      # common (AAF 0.6 - 0.9) and rare-minor (AAF 0.99+) variants among them
    bs, n = G.shape
    if route == "int":                                         # some fractional dosages (units of 1 / 255) beside the whole ones
        rng = np.random.default_rng(5)
        pick = (rng.random(G.shape) < 0.004) & ~np.isnan(G)
        v = np.clip(np.nan_to_num(G) * 255 + rng.integers(-40, 41, G.shape), 0, 510)
        G = np.where(pick, v / 255.0, G)
    P = y.shape[1]
    nulls, fo = _nulls(X, off, y, mask)
    fitted = np.array([nl["p"] for nl in nulls])
    with Step2QT(n, X.shape[1], P) as s2:
        s2.set_sparse_rule(n, 0.5, False)
        s2.bt_set_null(X.T, y.T, mask.T, fitted, firth_offset=fo)
        if route == "packed":
            got = s2.bt_score_packed(_pack(G))
        else:
            got = s2.bt_score_int(np.where(np.isnan(G), 0xFFFF, np.rint(np.nan_to_num(G) * 255)).astype(np.uint16), 255)
        gi, sg, pairs = [], [], []
        nflip = nflip_sparse = 0
        for j in range(bs):
            gk, flipped = bt.flip_geno(np.where(np.isnan(G[j]), -3.0, G[j]))
            assert flipped == (j % 2 == 0)
            obs = gk >= 0
            g = np.where(obs, gk, gk[obs].mean())              # the coding regenie tests, mean-imputed
            sgn = -1.0 if flipped else 1.0
            gi.append(g); sg.append(sgn)
            assert got["mean"][j] == pytest.approx(np.nanmean(G[j]), rel=1e-12)
            sparse = int((g != 0).sum()) <= 0.5 * n
            assert bool(got["sparse"][j]) == sparse, (j, flipped)
            nflip += flipped
            nflip_sparse += flipped and sparse
            for q in range(P):
                ref = bt.score_bt(g, X, y[:, q], mask[:, q].astype(float), nulls[q], sparse=sparse)
                assert ref is not None and not got["test_ignored"][j, q]
                assert got["stats"][j, q] == pytest.approx(sgn * ref["stats"], rel=1e-9, abs=1e-10)
                assert got["bhat"][j, q] == pytest.approx(sgn * ref["bhat"], rel=1e-9, abs=1e-12)
                assert got["denum"][j, q] == pytest.approx(ref["denum"], rel=1e-9)
                pairs.append((j, q, ref))
        assert nflip == bs // 2 and nflip_sparse >= bs // 4           # the flipped rare variants, and the flipped common ones below 29 % MAF
        var = np.array([p[0] for p in pairs], np.int32)
        tr = np.array([p[1] for p in pairs], np.int32)
        fast = np.array([int(got["sparse"][p[0]]) for p in pairs], np.uint8)
        fc = s2.bt_correct(BT_FIRTH_APPROX, var, tr, fast)
        sc = s2.bt_correct(BT_SPA, var, tr, fast)
    nfirth = nspa = 0
    for t, (j, q, ref) in enumerate(pairs):
        g, m, sgn = gi[j], mask[:, q].astype(float), sg[j]
        is_fast = bool(fast[t])
        want = bt.approx_firth(g, X, y[:, q], m, nulls[q], fo[q], sparse=is_fast, mac=0 if is_fast else None)
        if want is None:
            assert fc["fail"][t] == 1
        else:
            assert fc["fail"][t] == 0
            assert fc["beta"][t] == pytest.approx(sgn * want["bhat"], rel=1e-6, abs=1e-8)
            assert fc["se"][t] == pytest.approx(want["se"], rel=1e-6)
            assert fc["chisq"][t] == pytest.approx(want["chisq"], rel=1e-6, abs=1e-9)
            nfirth += 1
        if abs(ref["stats"]) < 0.1:          # (a saddlepoint at t ~ 0, p ~ 1: -log10 p is all cancellation there, and regenie corrects |z| above its threshold only)
            continue
        wsp = bt.spa_test(ref["stats"], ref["denum"], ref["Gres"], nulls[q], m, carriers=np.flatnonzero(g != 0) if is_fast else None)
        if wsp is None:
            assert sc["fail"][t] == 1
        else:
            assert sc["fail"][t] == 0
            assert sc["logp"][t] == pytest.approx(wsp["logp"], rel=1e-6, abs=1e-9)
            assert sc["chisq"][t] == pytest.approx(wsp["chisq"], rel=1e-6, abs=1e-9)
            assert sc["beta"][t] == pytest.approx(sgn * wsp["bhat"], rel=1e-6, abs=1e-10) and sc["se"][t] == pytest.approx(wsp["se"], rel=1e-9)
            nspa += 1
    assert nfirth > len(pairs) // 2 and nspa > len(pairs) // 2


def test_bt_usage_errors():
    from regenie_amd.engine import RgError
    from regenie_amd.step2 import BT_FIRTH_APPROX, Step2QT
    X, off, G, y, mask = _problem(3, n=801, bs=8)
    n, P = G.shape[1], y.shape[1]
    with Step2QT(n, X.shape[1], P) as s2:
        with pytest.raises(RgError):
            s2.bt_score_packed(_pack(G))                                   # no null model
        nulls, fo = _nulls(X, off, y, mask)
        s2.bt_set_null(X.T, y.T, mask.T, np.array([nl["p"] for nl in nulls]))
        with pytest.raises(RgError):
            s2.bt_correct(BT_FIRTH_APPROX, [0], [0], [0])                  # no block scored
        s2.bt_score_packed(_pack(G))
        with pytest.raises(RgError):
            s2.bt_correct(BT_FIRTH_APPROX, [0], [0], [0])                  # null model without firth_offset
        with pytest.raises(RgError):
            s2.bt_correct(2, [99], [0], [0])                               # pair out of range
