"""Step-2 QT score test on the GPU (include/rg_step2.h through regenie_amd.step2.Step2QT) against the CPU restatement of
the reference (oracle/regenie_step2_qt.py: score_qt_block_ref makes the reference's per-variant choice between the sparse and the
dense branch of compute_score_qt).  fp64 throughout; tolerance 1e-9 relative (the kernel sums 2048-sample chunk
partials in chunk order, the reference sums along Eigen's vectorised order)."""
import numpy as np
import pytest

from oracle import regenie_step2_qt as s2o

pytestmark = pytest.mark.gpu

RTOL = 1e-9


def _problem(seed, n, C, P, bs, miss_y=0.05):
    rng = np.random.default_rng(seed)
    cov = np.column_stack([np.ones(n), rng.normal(size=(n, C - 1))]) if C > 1 else np.ones((n, 1))
    X = np.linalg.qr(cov)[0]
    Y = rng.normal(size=(n, P)) + 0.3 * rng.normal(size=(n, 1))
    mask = np.ones((n, P))
    if miss_y:
        mask[rng.random((n, P)) < miss_y] = 0
    Y = (Y - X @ (X.T @ Y)) * mask
    neff = mask.sum(axis=0)
    scale_Y = np.sqrt((Y ** 2).sum(axis=0) / (neff - C))
    Y = Y / scale_Y
    blup = 0.1 * rng.normal(size=(n, P)) * mask
    res, _, scf = s2o.compute_res(Y, blup, mask, neff, C, scale_Y)
    G = rng.binomial(2, rng.uniform(0.02, 0.5, size=bs)[:, None], size=(bs, n)).astype(np.float64)
    return X, res, mask, scf, G


def _run(X, res, mask, scf, G, **kw):
    from regenie_amd.step2 import Step2QT
    n, C = X.shape
    with Step2QT(n, C, res.shape[1]) as s2:
        s2.set_null(X.T, res.T, mask.T, scf)
        return s2.score_block(G, **kw)


def _compare(got, ref):
    assert np.array_equal(got["ignored"], ref["ignored"])
    assert np.array_equal(got["n_obs"], ref["n_obs"])
    ok = ref["ignored"] == 0
    assert np.isnan(got["stats"][~ok]).all() and np.isnan(got["bhat"][~ok]).all()
    for k in ("stats", "bhat", "se", "chisq"):
        scale = np.nanmax(np.abs(ref[k][ok]))
        assert np.allclose(got[k][ok], ref[k][ok], rtol=RTOL, atol=RTOL * scale, equal_nan=True), k   # NaN: an all-zero variant on the sparse branch (0 / 0)
    obs = ref["n_obs"] > 0
    assert np.allclose(got["mean"][obs], ref["mean"][obs], rtol=1e-13)
    assert np.allclose(got["scale_fac"][ok], ref["scale_fac"][ok], rtol=1e-11)


@pytest.mark.parametrize("n,C,P,bs", [(5003, 5, 3, 37), (300, 1, 1, 1), (2048, 3, 2, 4), (4097, 12, 7, 9)])
def test_parity_with_oracle(n, C, P, bs):
    X, res, mask, scf, G = _problem(n + bs, n, C, P, bs)
    rng = np.random.default_rng(7)
    if bs >= 4:
        G[0, rng.random(n) < 0.1] = np.nan            # missing as NaN
        G[1, rng.random(n) < 0.02] = -3.0             # missing as regenie's code
        G[2, :] = 2.0                                 # monomorphic -> scale_fac < numtol -> ignored
        G[3, :] = np.nan                              # nothing observed -> ignored
    G = G + (rng.random(G.shape) < 0.3) * rng.uniform(0, 0.01, size=G.shape) * (G < 1.5)   # dosages, not just hardcalls
    if bs >= 4:
        G[2, :] = 2.0
    ref = s2o.score_qt_block_ref(G, X, res, mask, scf)
    assert bs <= 4 or 0 < ref["sparse"].sum() < bs           # both branches of compute_score_qt (5 % of the phenotype values are missing)
    _compare(_run(X, res, mask, scf, G), ref)


@pytest.mark.parametrize("tile", ["4x4", "4x8", "8x4", "8x8", "16x4"])
def test_tiles_agree_with_oracle(tile, monkeypatch):
    """RG_S2_TILE picks the workgroup tile of the two streaming kernels; every tile must pass the same parity bar."""
    monkeypatch.setenv("RG_S2_TILE", tile)
    X, res, mask, scf, G = _problem(21, 10_007, 6, 5, 43)
    G[0, ::9] = np.nan
    G[7, :] = 0.0
    _compare(_run(X, res, mask, scf, G), s2o.score_qt_block_ref(G, X, res, mask, scf))
    monkeypatch.setenv("RG_S2_TILE", "3x3")
    from regenie_amd.engine import RgError
    with pytest.raises(RgError, match="RG_S2_TILE"):
        _run(X, res, mask, scf, G)


def test_device_pointer_input_and_determinism():
    import torch
    X, res, mask, scf, G = _problem(11, 9000, 6, 4, 50)
    G[5, ::11] = np.nan
    a = _run(X, res, mask, scf, G)
    big = torch.full((50, 9216), float("nan"), dtype=torch.float64, device="cuda")   # padded leading dimension
    big[:, :9000] = torch.from_numpy(G).cuda()
    b = _run(X, res, mask, scf, big[:, :9000])
    c = _run(X, res, mask, scf, G)
    for k in ("stats", "bhat", "scale_fac", "mean"):
        assert np.array_equal(a[k], b[k], equal_nan=True), k
        assert np.array_equal(a[k], c[k], equal_nan=True), k


def test_invariances_at_scale():
    """Sizes the oracle does not loop over: the statistic ignores covariate-space shifts and the genotype scale, bhat
    scales inversely, and a sub-block equals the same rows of the full block bit for bit."""
    n, C, P, bs = 120_000, 10, 10, 256
    X, res, mask, scf, G = _problem(3, n, C, P, bs, miss_y=0)     # complete phenotypes: the sparse and the dense branch are one number
    rng = np.random.default_rng(5)
    base = _run(X, res, mask, scf, G)
    assert not base["ignored"].any() and np.isfinite(base["stats"]).all()
    shifted = _run(X, res, mask, scf, 0.5 * G + 3.0 + (X[:, 1:] @ rng.normal(size=(C - 1, bs))).T * 5.0 + 40.0)
    assert np.allclose(shifted["stats"], base["stats"], rtol=1e-7, atol=1e-9)
    assert np.allclose(shifted["bhat"], 2.0 * base["bhat"], rtol=1e-7, atol=1e-12)
    sub = _run(X, res, mask, scf, G[64:131])
    assert np.array_equal(sub["stats"], base["stats"][64:131]) and np.array_equal(sub["bhat"], base["bhat"][64:131])
    # spot check against the oracle on a few rows
    ref = s2o.score_qt_block_ref(G[:6], X, res, mask, scf)
    assert np.allclose(base["stats"][:6], ref["stats"], rtol=RTOL, atol=1e-10)
    assert np.allclose(base["bhat"][:6], ref["bhat"], rtol=RTOL, atol=1e-13)


def test_argument_errors():
    from regenie_amd.engine import RgError
    from regenie_amd.step2 import Step2QT
    with pytest.raises(RgError):
        Step2QT(100, 65, 1)
    with pytest.raises(RgError):
        Step2QT(10, 10, 1)
    with Step2QT(100, 2, 1) as s2:
        with pytest.raises(RgError, match="set_null"):
            s2.score_block(np.zeros((3, 100)))


def test_step1_loco_feeds_step2(example_dir, tmp_path):
    """The reference's two-step flow on its own example: `--step 1` (the C++ driver of this repo) writes the LOCO files, the
    chromosome-2 row becomes the `blup` of compute_res (Data.cpp:2386-2400), and the chromosome-2 variants of the .bed go
    through the score test -- GPU against the oracle on identical inputs."""
    import os
    import subprocess
    from oracle import regenie_step1 as orc
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    E = example_dir
    common = dict(bed=os.path.join(E, "example_3chr"), pheno_file=os.path.join(E, "phenotype.txt"),
                  covar_file=os.path.join(E, "covariates.txt"), bsize=100)
    r = subprocess.run([os.path.join(root, "regenie_amd", "bin", "regenie-amd"), "--step", "1", "--bed", common["bed"],
                        "--phenoFile", common["pheno_file"], "--covarFile", common["covar_file"], "--bsize", "100",
                        "--out", str(tmp_path / "s1")], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    bim, chrom, offs, snp_ids, prep = orc.load_inputs(orc.Step1Options(out=str(tmp_path / "unused"), **common))
    ia = prep.ind_in_analysis
    ids = [i for i, k in zip(prep.ids, ia) if k]
    n, P = int(ia.sum()), prep.Y.shape[1]
    blup = np.zeros((n, P))
    for k in range(P):
        lines = open(str(tmp_path / ("s1_%d.loco" % (k + 1)))).read().split("\n")
        hdr = lines[0].split(" ")[1:-1]             # the writer's own sample order (Data.cpp:1934); Step 2 matches by name
        assert sorted(hdr) == sorted(ids)
        row = [ln for ln in lines[1:-1] if ln.split(" ")[0] == "2"][0]
        val = dict(zip(hdr, (float(v) for v in row.split(" ")[1:-1])))
        blup[:, k] = [val[i] for i in ids]
    X, Y, mask = prep.X[ia], prep.Y[ia], prep.mask[ia].astype(np.float64)
    res, _, scf = s2o.compute_res(Y, blup * mask, mask, prep.Neff, X.shape[1], prep.scale_Y)
    rows, _ = orc.open_bed(common["bed"] + ".bed", prep.n_file)
    sel = np.flatnonzero(chrom == 2)
    G = orc.decode_bed_rows(np.asarray(rows[sel]), prep.n_file)[:, ~prep.ind_ignore][:, ia]
    assert G.shape == (len(sel), n) and len(sel) > 50
    ref = s2o.score_qt_block_ref(G, X, res, mask, scf)
    assert (ref["ignored"] == 0).sum() > 50 and np.nanmax(np.abs(ref["stats"])) > 1.0
    _compare(_run(X, res, mask, scf, G), ref)


# ---- hard calls: rg_s2_qt_block_packed (2-bit rows, exact i8 matrix-core contractions) -------------------------------------------
def _pack_bed(G):
    """[bs][n] dosages in {0, 1, 2, nan} -> .bed rows (00 -> 2, 01 -> missing, 10 -> 1, 11 -> 0; low bits first; the unused bits
    of the last byte are 00, as plink writes them)."""
    code = np.where(np.isnan(G), 1, np.where(G == 2, 0, np.where(G == 1, 2, 3))).astype(np.uint8)
    bs, n = code.shape
    pad = (-n) % 4
    if pad:
        code = np.concatenate([code, np.zeros((bs, pad), np.uint8)], axis=1)
    c = code.reshape(bs, -1, 4)
    return (c[:, :, 0] | (c[:, :, 1] << 2) | (c[:, :, 2] << 4) | (c[:, :, 3] << 6)).astype(np.uint8)


def _run_packed(X, res, mask, scf, rows, **kw):
    from regenie_amd.step2 import Step2QT
    n, C = X.shape
    with Step2QT(n, C, res.shape[1]) as s2:
        s2.set_null(X.T, res.T, mask.T, scf)
        return s2.score_block_packed(rows, **kw)


@pytest.mark.parametrize("n,C,P,bs", [(5003, 5, 3, 37), (301, 1, 1, 5), (2048, 3, 2, 4), (4097, 12, 7, 9), (70_001, 11, 9, 300), (9_999, 3, 2, 1)])
def test_packed_parity_with_oracle(n, C, P, bs):
    """Tail samples (n not a multiple of 4 nor of 64), more than 16 contraction columns (two column groups), several sample
    segments, more than one 128-row tile, missing calls, a monomorphic and an all-missing variant."""
    X, res, mask, scf, G = _problem(n + bs, n, C, P, bs, miss_y=0)
    rng = np.random.default_rng(9)
    if bs >= 4:
        G[0, rng.random(n) < 0.1] = np.nan
        G[1, rng.random(n) < 0.002] = np.nan
        G[2, :] = 2.0
        G[3, :] = np.nan
    ref = s2o.score_qt_block_ref(G, X, res, mask, scf)
    assert 0 < ref["sparse"].sum() < bs or bs < 4
    got = _run_packed(X, res, mask, scf, _pack_bed(G))
    _compare(got, ref)
    dense = _run(X, res, mask, scf, G)
    ok = ref["ignored"] == 0
    assert np.allclose(got["stats"][ok], dense["stats"][ok], rtol=1e-10, atol=1e-11)     # mask all ones: both branches are one number


def test_packed_no_missing_calls_and_flip():
    """No missing call in the block (the missing-indicator contraction is skipped), and --ref-first = 2 - g on the observed calls."""
    X, res, mask, scf, G = _problem(77, 12_345, 4, 3, 200, miss_y=0)
    ref = s2o.score_qt_block_ref(G, X, res, mask, scf)
    _compare(_run_packed(X, res, mask, scf, _pack_bed(G)), ref)
    G[10, ::7] = np.nan
    G[150, 5] = np.nan
    ref2 = s2o.score_qt_block_ref(2.0 - G, X, res, mask, scf)
    _compare(_run_packed(X, res, mask, scf, _pack_bed(G), flip=True), ref2)


def test_packed_device_rows_padded_ld_and_determinism():
    import torch
    X, res, mask, scf, G = _problem(13, 30_011, 6, 4, 260, miss_y=0)
    G[5, ::11] = np.nan
    rows = _pack_bed(G)
    a = _run_packed(X, res, mask, scf, rows)
    big = torch.randint(0, 256, (260, rows.shape[1] + 37), dtype=torch.uint8, device="cuda")   # garbage beyond the row
    big[:, :rows.shape[1]] = torch.from_numpy(rows).cuda()
    b = _run_packed(X, res, mask, scf, big)
    c = _run_packed(X, res, mask, scf, rows[100:200])
    for k in ("stats", "bhat", "scale_fac", "mean", "n_obs"):
        assert np.array_equal(a[k], b[k], equal_nan=True), k
        assert np.array_equal(a[k][100:200], c[k], equal_nan=True), k


@pytest.mark.parametrize("n,C,P,bs,n_samples", [(5003, 5, 3, 64, 5003), (2500, 1, 1, 9, 2500), (4097, 12, 7, 40, 4600), (30_011, 3, 6, 200, 31_000)])
def test_packed_masked_phenotypes_follow_the_reference_branches(n, C, P, bs, n_samples):
    """Phenotypes that differ in their missing values: per variant the reference's choice between the sparse branch of
    compute_score_qt (approximate per-trait denominators) and the dense one (oracle score_qt_block_ref, pinned against regenie's own
    output in tests/test_reference_pin.py); n_samples > n moves check_sparse_G's threshold.  Also the per-trait allele counts."""
    X, res, mask, scf, G = _problem(n + bs + 1, n, C, P, bs, miss_y=0.07)
    rng = np.random.default_rng(4)
    G[0, rng.random(n) < 0.1] = np.nan
    G[1, rng.random(n) < 0.002] = np.nan
    G[2, :] = 2.0
    G[3, :] = np.nan
    ref = s2o.score_qt_block_ref(G, X, res, mask, scf, n_samples=n_samples)
    dense = s2o.score_qt_block(G, X, res, mask, scf)
    from regenie_amd.step2 import Step2QT
    with Step2QT(n, C, P) as s2:
        s2.set_null(X.T, res.T, mask.T, scf)
        s2.set_sparse_rule(n_samples)
        got = s2.score_block_packed(_pack_bed(G))
        again = s2.score_block_packed(_pack_bed(G)[5:9])
    _compare(got, ref)
    with Step2QT(n, C, P) as s2:                                   # the fp64 (dosage) route makes the same per-variant choice
        s2.set_null(X.T, res.T, mask.T, scf)
        s2.set_sparse_rule(n_samples)
        _compare(s2.score_block(G), ref)
    ok = ref["ignored"] == 0
    sp = (ref["sparse"] == 1) & ok
    assert sp.sum() > 0 and (~sp & ok).sum() > 0
    assert np.abs(ref["chisq"][sp] / dense["chisq"][sp] - 1.0).max() > 1e-4          # the branches are different numbers here
    obs = ~np.isnan(G)
    assert np.array_equal(got["n_obs_p"], (obs[:, :, None] & (mask[None] > 0)).sum(axis=1))
    assert np.array_equal(got["total_p"], np.einsum("jn,np->jp", np.where(obs, G, 0.0), mask))
    assert np.array_equal(again["stats"], got["stats"][5:9], equal_nan=True)


@pytest.mark.parametrize("n,C,P,bs,miss_y", [(5003, 5, 3, 64, 0.07), (4097, 12, 7, 40, 0.05), (6001, 3, 40, 33, 0.02), (3001, 17, 6, 20, 0.1), (4000, 4, 5, 30, 0.4)])
def test_packed_masked_compact_axis_equals_mask_columns(n, C, P, bs, miss_y, monkeypatch):
    """Round 6: for phenotypes that differ in their missing values the hard-call route takes the sums against the mask columns as
    (all samples) - (the samples masked for the trait) over a compact sample axis (k_s2_compact_rows / k_s2_count_traits / k_s2_combine_traits)
    whenever the masked-sample lists hold at most n entries in all; RG_S2_MASK_COLS=1 keeps rounds 4 - 5's C P + P mask columns.  Both against
    each other and against the oracle: 40 traits need two contraction launches of 32 + 8, 17 covariates two column groups, 40 % missing values in
    five traits exceed n entries (the library itself falls back on the mask columns: both runs are then the same route)."""
    X, res, mask, scf, G = _problem(1234 + P, n, C, P, bs, miss_y=miss_y)
    rng = np.random.default_rng(11)
    G[rng.random(G.shape) < 0.01] = np.nan
    G[1, :] = np.nan
    rows = _pack_bed(G)
    from regenie_amd.step2 import Step2QT
    out = {}
    for name, env in (("compact", None), ("columns", "1")):
        if env is None:
            monkeypatch.delenv("RG_S2_MASK_COLS", raising=False)
        else:
            monkeypatch.setenv("RG_S2_MASK_COLS", env)
        with Step2QT(n, C, P) as s2:
            s2.set_null(X.T, res.T, mask.T, scf)
            out[name] = s2.score_block_packed(rows)
            out[name + "_again"] = s2.score_block_packed(rows[3:11])
    a, b = out["compact"], out["columns"]
    for k in ("n_obs", "ignored", "n_obs_p", "total_p"):
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    for k in ("stats", "bhat"):
        ok = ~np.isnan(b[k])
        assert np.array_equal(np.isnan(a[k]), np.isnan(b[k])), k
        assert np.allclose(a[k][ok], b[k][ok], rtol=1e-10, atol=1e-12), (k, np.abs(a[k][ok] - b[k][ok]).max())
    assert np.array_equal(out["compact_again"]["stats"], a["stats"][3:11], equal_nan=True)
    _compare(a, s2o.score_qt_block_ref(G, X, res, mask, scf))


def test_packed_planes_follow_set_null():
    """A second rg_s2_set_null with other residuals (next chromosome) and then with other masks: the cached digit planes are rebuilt
    for exactly what changed."""
    from regenie_amd.step2 import Step2QT
    n, C, P, bs = 6000, 4, 3, 50
    X, res, mask, scf, G = _problem(91, n, C, P, bs, miss_y=0.05)
    _, res2, _, scf2, _ = _problem(92, n, C, P, bs, miss_y=0.05)
    res2 = res2 * mask
    mask3 = mask.copy(); mask3[::5, 1] = 0
    rows = _pack_bed(G)
    with Step2QT(n, C, P) as s2:
        for (r_, m_, f_) in ((res, mask, scf), (res2, mask, scf2), (res2 * mask3, mask3, scf2), (res * 0 + res2, np.ones_like(mask), scf)):
            s2.set_null(X.T, r_.T, m_.T, f_)
            _compare(s2.score_block_packed(rows), s2o.score_qt_block_ref(G, X, r_, m_, f_))


def test_packed_at_scale_matches_dense_route():
    """200,000 samples x 1,024 variants (BASELINE configs[4] has 500,000 samples: that size is checked against the oracle by bench.py's `step2` record, tools/step2_record.py): the packed route against the fp64 route of the
    same library, and a spot check against the oracle."""
    n, C, P, bs = 200_000, 10, 10, 1024
    X, res, mask, scf, G = _problem(31, n, C, P, bs, miss_y=0)
    rng = np.random.default_rng(2)
    G[rng.random(G.shape) < 0.01] = np.nan
    rows = _pack_bed(G)
    got = _run_packed(X, res, mask, scf, rows)
    dense = _run(X, res, mask, scf, G)
    assert not got["ignored"].any()
    assert np.array_equal(got["n_obs"], dense["n_obs"])
    assert np.allclose(got["stats"], dense["stats"], rtol=1e-9, atol=1e-10)
    assert np.allclose(got["bhat"], dense["bhat"], rtol=1e-9, atol=1e-13)
    ref = s2o.score_qt_block_ref(G[:4], X, res, mask, scf)
    assert np.allclose(got["stats"][:4], ref["stats"], rtol=RTOL, atol=1e-10)
    print("packed kernel %.3f ms, dense kernel %.3f ms for %d variants x %d samples" % (got["kernel_ms"], dense["kernel_ms"], bs, n))


def test_contraction_primitive_matches_numpy():
    """rg_s2_set_columns / rg_s2_contract_packed: hard-call rows against arbitrary fp64 columns (three column groups, wide dynamic
    range inside a column, an all-zero column), the g^2 contraction of the first columns, the call counts; with and without flip."""
    from regenie_amd.step2 import Step2QT
    rng = np.random.default_rng(8)
    n, ncol, nsq, bs = 20_003, 37, 5, 130
    cols = rng.normal(size=(ncol, n)) * np.exp(rng.uniform(-12, 12, size=(ncol, 1)))
    cols[3] *= np.exp(rng.uniform(-20, 0, size=n))           # entries far below the column's largest: truncated at 2^-54 of it
    cols[7] = 0.0
    cols[0] = rng.random(n)                                    # a weight column, as the binary-trait test uses
    G = rng.binomial(2, rng.uniform(0.02, 0.5, size=bs)[:, None], size=(bs, n)).astype(np.float64)
    G[rng.random(G.shape) < 0.01] = np.nan
    G[5] = np.where(np.isnan(G[5]), 0.0, G[5])                 # a row without missing calls
    rows = _pack_bed(G)
    with Step2QT(n, 2, 1) as s2:
        s2.set_columns(cols, n_sq=nsq)
        for flip in (False, True):
            g = 2.0 - G if flip else G
            g0, miss = np.where(np.isnan(g), 0.0, g), np.isnan(g).astype(np.float64)
            got = s2.contract_packed(rows, flip=flip)
            assert np.array_equal(got["counts"][:, 0], (g0 == 1).sum(axis=1)) and np.array_equal(got["counts"][:, 1], (g0 == 2).sum(axis=1))
            assert np.array_equal(got["counts"][:, 2], miss.sum(axis=1).astype(np.int32))
            for name, want, have in (("g0", g0 @ cols.T, got["sums"][:, 0]), ("miss", miss @ cols.T, got["sums"][:, 1]),
                                     ("sq", (g0 * g0) @ cols[:nsq].T, got["sq"])):
                bound = (np.abs(g0) + miss) @ np.abs(cols[: want.shape[1]]).T + 1e-300     # sum of the terms' magnitudes
                assert (np.abs(have - want) <= 4e-15 * bound + 2.0 ** -52 * n * np.abs(cols[: want.shape[1]]).max(axis=1)[None, :]).all(), name


@pytest.mark.parametrize("scale,n,C,P,bs,miss_y", [(255, 5003, 5, 3, 64, 0.0), (255, 9001, 12, 7, 130, 0.06), (16384, 4097, 3, 2, 40, 0.05),
                                                    (1, 3000, 2, 2, 20, 0.04), (4063, 2500, 1, 1, 9, 0.0), (4064, 2500, 1, 1, 9, 0.0)])
def test_integer_dosage_route_against_oracle(scale, n, C, P, bs, miss_y):
    """rg_s2_qt_block_int: dosages that are integers in units of 1 / scale (8-bit .bgen probabilities, .pgen's 16-bit dosages, hard
    calls as scale 1; two digit planes up to scale 4063, three above) against the reference-pinned oracle on the same values as doubles,
    and against the library's fp64 route; complete phenotypes and phenotypes that differ in their missing values (masked-sample lists)."""
    from regenie_amd.step2 import Step2QT
    X, res, mask, scf, _ = _problem(n + bs + scale, n, C, P, bs, miss_y=miss_y)
    rng = np.random.default_rng(scale + n)
    af = rng.uniform(0.01, 0.6, size=(bs, 1))
    hard = rng.binomial(2, af, size=(bs, n))
    Gi = np.clip(hard * scale + (rng.random((bs, n)) < 0.4) * rng.integers(-scale // 3, scale // 3 + 1, size=(bs, n)), 0, 2 * scale).astype(np.int64)
    Gi[rng.random((bs, n)) < 0.3] = 0                                # plenty of exact zeros: both branches of check_sparse_G
    miss = rng.random((bs, n)) < 0.004
    miss[0] = rng.random(n) < 0.2
    miss[3] = True                                                   # nothing observed
    Gi[2] = 2 * scale                                                # monomorphic
    miss[2] = False
    Gi[4] = 2 * scale - (np.arange(n) % 3 == 0) * max(1, scale // 8)  # the largest values (third digit at scale >= 4064); not a near-constant row:
                                                                     # |r|^2 is formed as sum g~^2 - |beta|^2, ill-conditioned only below the MAC filter
    G = np.where(miss, np.nan, Gi / float(scale))
    ref = s2o.score_qt_block_ref(G, X, res, mask, scf)
    assert 0 < ref["sparse"].sum() < bs
    with Step2QT(n, C, P) as s2:
        s2.set_null(X.T, res.T, mask.T, scf)
        got = s2.score_block_int(np.where(miss, 0xFFFF, Gi).astype(np.uint16), scale)
        dense = s2.score_block(G)
    _compare(got, ref)
    ok = ref["ignored"] == 0
    assert np.allclose(got["stats"][ok], dense["stats"][ok], rtol=1e-9, atol=1e-10, equal_nan=True)


@pytest.mark.parametrize("scale", [255, 16384])
def test_contraction_primitive_for_integer_dosages(scale):
    from regenie_amd.step2 import Step2QT
    rng = np.random.default_rng(scale)
    n, ncol, nsq, bs = 12_345, 21, 3, 70
    cols = rng.normal(size=(ncol, n)) * np.exp(rng.uniform(-8, 8, size=(ncol, 1)))
    cols[:nsq] = rng.random((nsq, n))
    Gi = rng.integers(0, 2 * scale + 1, size=(bs, n))
    Gi[rng.random((bs, n)) < 0.5] = 0
    miss = rng.random((bs, n)) < 0.01
    g0 = np.where(miss, 0.0, Gi / float(scale))
    with Step2QT(n, 2, 1) as s2:
        s2.set_columns(cols, n_sq=nsq)
        got = s2.contract_int(np.where(miss, 0xFFFF, Gi).astype(np.uint16), scale)
    assert np.array_equal(got["vstat"][:, 0], np.where(miss, 0, Gi).sum(axis=1)) and np.array_equal(got["vstat"][:, 1], (np.where(miss, 0, Gi) ** 2).sum(axis=1))
    assert np.array_equal(got["vstat"][:, 2], (~miss).sum(axis=1)) and np.array_equal(got["vstat"][:, 3], ((Gi != 0) & ~miss).sum(axis=1))
    for name, want, have in (("g0", g0 @ cols.T, got["sums"][:, 0]), ("miss", miss.astype(float) @ cols.T, got["sums"][:, 1]), ("sq", (g0 * g0) @ cols[:nsq].T, got["sq"])):
        bound = (np.abs(g0) + miss) @ np.abs(cols[: want.shape[1]]).T * (2.0 if name == "sq" else 1.0) + 1e-300
        assert (np.abs(have - want) <= 1e-13 * bound).all(), name
