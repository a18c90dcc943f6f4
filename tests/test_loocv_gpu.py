"""LOOCV and binary-trait paths on the GPU against the oracle (through the C ABI).

The BT case is the reference's own Step-1 test command (test/test_bash.sh:62-80): N = 494 < 5000 forces
LOOCV (Data.cpp:353-356), and its log must carry `0.4504` on the `min value` line."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
pytest.importorskip("torch")

from oracle import regenie_step1 as orc  # noqa: E402
from regenie_amd.engine import Step1Engine  # noqa: E402
from tests.util import rel_err, synth_dosages, write_plink  # noqa: E402


def _level0_loocv(opt):
    """Level-0 LOOCV predictors of every block through the C ABI; returns (ref Step1Result, W_gpu list)."""
    ref = orc.run_step1(opt)
    assert ref.use_loocv
    bim, chrom, offs, snp_ids, prep = orc.load_inputs(opt)
    bed, _ = orc.open_bed(opt.bed + ".bed", prep.n_file)
    B = len(ref.blocks)
    eng = Step1Engine(0)
    eng.set_problem(X=prep.X, Y=prep.Y, mask=prep.mask, ind_in_analysis=prep.ind_in_analysis, cv_sizes=None,
                    lam=ref.lam, neff=prep.Neff, n_file=prep.n_file, n_blocks_total=B, max_block_size=opt.bsize,
                    ind_ignore=prep.ind_ignore if prep.ind_ignore.any() else None, ref_first=opt.ref_first)
    rows = [np.ascontiguousarray(bed[offs[s:s + bs]]) for (_, s, bs) in ref.blocks]
    eng.l0_blocks_host(list(range(B)), rows)
    eng.sync()
    N, P = prep.Y.shape
    R0 = ref.lam.size
    W = [np.zeros((N, B * R0)) for _ in range(P)]
    for b in range(B):
        for ph in range(P):
            W[ph][:, b * R0:(b + 1) * R0] = eng.get_w(b, ph)
    return ref, W, eng


def test_level0_loocv_reference_bt_command(example_dir):
    E = example_dir
    opt = orc.Step1Options(bed=os.path.join(E, "example"), pheno_file=os.path.join(E, "phenotype_bin.txt"),
                           covar_file=os.path.join(E, "covariates.txt"), remove=[os.path.join(E, "fid_iid_to_remove.txt")],
                           exclude=[os.path.join(E, "snplist_rm.txt")], bsize=100, bt=True)
    ref, W, eng = _level0_loocv(opt)
    eng.close()
    for ph in range(2):
        assert rel_err(W[ph], ref.W[ph]) < 1e-8


def test_level0_loocv_qt_missing(tmp_path):
    N, M = 700, 300
    g = synth_dosages(M, N, miss_rate=0.02, seed=33)
    pre = str(tmp_path / "lo")
    write_plink(pre, g, np.repeat([1, 3], [150, 150]), P=2, ncov=2, seed=6)
    opt = orc.Step1Options(bed=pre, pheno_file=pre + ".pheno", covar_file=pre + ".covar", bsize=70, loocv=True)
    ref, W, eng = _level0_loocv(opt)
    eng.close()
    for ph in range(2):
        assert rel_err(W[ph], ref.W[ph]) < 1e-8


def test_level0_loocv_more_samples_than_one_chunk(tmp_path):
    """70,000 samples: the leave-one-out level 0 (loocv_tri.hip) transforms the genotypes 65,536 sample positions at a time (Z = Q^T G~ per chunk,
    the recurrences per chunk) -- two chunks here, the second a partial one -- with missing calls and a sample filter; blocks of 48 and 16 SNPs
    (orders that are not multiples of the 64-wide tiles)."""
    N, M = 70000, 64
    g = synth_dosages(M, N, miss_rate=0.01, seed=41)
    pre = str(tmp_path / "big")
    write_plink(pre, g, np.repeat([1, 2], [48, 16]), P=2, ncov=2, seed=9)
    opt = orc.Step1Options(bed=pre, pheno_file=pre + ".pheno", covar_file=pre + ".covar", bsize=48, loocv=True)
    ref, W, eng = _level0_loocv(opt)
    eng.close()
    for ph in range(2):
        assert rel_err(W[ph], ref.W[ph]) < 1e-8


def _fast_dosages(M, N, miss_rate, seed):
    """Binomial(2, maf) hard calls with missing calls as -1 (numpy's generator: synth_dosages' per-element hash takes 20 s at this size)."""
    rng = np.random.default_rng(seed)
    maf = rng.uniform(0.05, 0.5, (M, 1))
    g = rng.binomial(2, maf, (M, N)).astype(np.int8)
    g[rng.random((M, N)) < miss_rate] = -1
    return g


@pytest.mark.parametrize("binary", [False, True], ids=["qt", "bt"])
def test_level0_loocv_at_the_block_order_it_is_benchmarked_at(tmp_path, binary):
    """Block orders 1,024 / 1,000 / 960 (a full last tile, BASELINE's bsize with its padded tile, a tile edge) x 20,000 samples: the tridiagonal
    leave-one-out level 0 (loocv_tri.hip: up to 1,023 Householder steps per block) against the oracle's eigendecomposition route
    (ridge_level_0_loocv, Step1_Models.cpp:615-726).  The long reflector chain is what the small-order tests cannot show."""
    N = 20000
    g = _fast_dosages(1024 + 1000 + 960, N, 0.01, 51 + binary)
    pre = str(tmp_path / "lobig")
    write_plink(pre, g, np.repeat([1, 2, 3], [1024, 1000, 960]), P=2, ncov=2, seed=11, binary=binary)
    opt = orc.Step1Options(bed=pre, pheno_file=pre + ".pheno", covar_file=pre + ".covar", bsize=1024, loocv=True, bt=binary)
    ref, W, eng = _level0_loocv(opt)
    eng.close()
    assert [b[2] for b in ref.blocks] == [1024, 1000, 960]
    for ph in range(2):
        assert rel_err(W[ph], ref.W[ph]) < 1e-8
        for b in range(3):      # per block too: one bad block must not hide behind the others' scale
            assert rel_err(W[ph][:, 5 * b:5 * b + 5], ref.W[ph][:, 5 * b:5 * b + 5]) < 1e-8
