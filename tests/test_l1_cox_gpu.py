"""Time-to-event traits behind the C ABI (include/rg_step1.h: rg_l1_cox; regenie_amd/csrc/l1x.hip "Cox ridge at level 1") against the
oracle (oracle/regenie_step1_t2e.py, pinned against regenie's own --t2e run by tests/test_reference_pin.py::test_t2e_cox_ridge_synthetic):
level 0 of the time columns on the GPU, then per trait the penalty grid, the held-out deviances, the selected penalty and the
out-of-fold predictions.  The library makes the coordinate pass of an IRLS iteration as a Gauss-Seidel sweep on the weighted Gram
(fp64 matrix cores) where the oracle passes over the samples: same arithmetic, another summation order."""
import numpy as np
import pytest

from oracle import regenie_step1 as orc
from oracle import regenie_step1_t2e as t2e
from tests.test_reference_pin import synth_t2e_case

pytestmark = pytest.mark.gpu


def _gpu_t2e(opt, t2e_map):
    from regenie_amd.engine import Step1Engine
    bim = orc.read_bim(opt.bed + ".bim", opt.nchrom)
    fam_ids = orc.read_fam(opt.bed + ".fam")
    prep = t2e.read_t2e(opt, t2e_map, fam_ids)
    t2e.prep_run_t2e(prep, t2e_map, opt)
    bed, _ = orc.open_bed(opt.bed + ".bed", prep.n_file)
    blocks = orc.chrom_blocks(bim.chrom, bim.chr_read, opt.bsize)
    B = len(blocks)
    h0 = orc.set_ridge_params(opt.n_ridge_l0)
    lam = bim.chrom.size * (1 - h0) / h0
    cv_sizes = orc.set_folds(prep.ind_in_analysis, opt.cv_folds)
    times = sorted(t2e_map)
    cols = [prep.pheno_names.index(t) for t in times]            # the library gets the time columns only
    eng = Step1Engine(0)
    eng.set_problem(X=prep.X, Y=prep.Y[:, cols], mask=prep.mask[:, cols], ind_in_analysis=prep.ind_in_analysis, cv_sizes=cv_sizes, lam=lam,
                    neff=prep.Neff[cols], n_file=prep.n_file, n_blocks_total=B, max_block_size=opt.bsize,
                    ind_ignore=prep.ind_ignore if prep.ind_ignore.any() else None, ref_first=opt.ref_first)
    rows = [np.ascontiguousarray(bed[bim.offset[s:s + bs]]) for (_, s, bs) in blocks]
    eng.l0_blocks_host(list(range(B)), rows)
    eng.sync()
    chrcols = orc.chr_columns(blocks, bim.chr_read, lam.size)
    out = {}
    for k, tn in enumerate(times):
        ti, ei = prep.pheno_names.index(tn), prep.pheno_names.index(t2e_map[tn])
        tau, dev, conv, best, pred = eng.l1_cox(k, prep.Y_raw[:, ti], prep.Y_raw[:, ei], prep.offset[:, ti], [nn for (_, _, nn) in chrcols],
                                                n_ridge_l1=opt.n_ridge_l1)
        out[tn] = dict(tau=tau, deviance=dev, converged=conv, best=best, loco=orc.loco_from_predictions(pred, chrcols, opt.nchrom))
    eng.close()
    return out, prep


def test_cox_ridge_level_1_against_the_oracle(tmp_path):
    meta, pre = synth_t2e_case(tmp_path)
    opt = orc.Step1Options(bed=pre, pheno_file=pre + ".t2e", covar_file=pre + ".covar", bsize=100)
    m = {"T1": "E1", "T2": "E2"}
    ref = t2e.run_step1_t2e(opt, m)
    got, prep = _gpu_t2e(opt, m)
    for tn in m:
        r, g = ref["traits"][tn], got[tn]
        assert g["converged"] and r["converged"]
        assert g["tau"] == pytest.approx(r["tau"], rel=1e-9)
        assert g["deviance"] == pytest.approx(r["deviance"], rel=1e-7)
        assert g["best"] == r["best"]
        ti = prep.pheno_names.index(tn)
        ok = prep.mask[:, ti]
        scale = np.abs(r["loco"][ok]).max()
        assert np.abs(g["loco"][ok] - r["loco"][ok]).max() <= 1e-7 * scale


def test_cox_usage_errors(tmp_path):
    from regenie_amd.engine import RgError, Step1Engine
    rng = np.random.default_rng(1)
    n = 600
    X = np.linalg.qr(rng.normal(size=(n, 2)))[0]
    eng = Step1Engine(0)
    eng.set_problem(X=X, Y=rng.normal(size=(n, 1)), mask=np.ones((n, 1), bool), ind_in_analysis=np.ones(n, bool), cv_sizes=np.array([120] * 5),
                    lam=np.array([10.0, 100.0]), neff=np.array([float(n)]), n_file=n, n_blocks_total=1, max_block_size=8)
    eng.set_w(0, 0, rng.normal(size=(n, 2)))
    t, e, o = rng.exponential(size=n), (rng.random(n) < 0.6).astype(float), np.zeros(n)
    with pytest.raises(RgError):
        eng.l1_cox(1, t, e, o, [2])                    # no such phenotype
    with pytest.raises(RgError):
        eng.l1_cox(0, t, e, o, [3])                    # columns per chromosome do not add up
    tau, dev, conv, best, pred = eng.l1_cox(0, t, e, o, [2])
    assert conv and tau[0] > tau[-1] > 0 and np.isfinite(dev).all() and pred.shape == (n, 1)
    eng.close()
