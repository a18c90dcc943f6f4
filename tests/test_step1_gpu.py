"""End-to-end parity of the HIP Step-1 path against the oracle, through the C ABI (GPU).

Tolerance: BASELINE.json asks for LOCO predictors within 1e-5 relative (max|gpu-ref| / max|ref|);
the fp64 path is expected to be ~1e-9 or better and the tests hold it to 1e-8."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
pytest.importorskip("torch")

from oracle import regenie_step1 as orc  # noqa: E402
from tests.util import gpu_step1, rel_err, synth_dosages, write_plink  # noqa: E402

TOL = 1e-8


def _compare(opt):
    ref = orc.run_step1(opt)
    got = gpu_step1(opt)
    P = ref.prep.Y.shape[1]
    report = {}
    for ph in range(P):
        report["W%d" % ph] = rel_err(got["W"][ph], ref.W[ph])
        report["cs%d" % ph] = rel_err(got["cumsum"][ph], ref.cumsum[ph][:5])
        report["pred%d" % ph] = rel_err(got["pred"][ph], ref.predictions[ph])
        report["loco%d" % ph] = rel_err(got["loco"][ph], ref.loco[ph])
    print(report)
    for ph in range(P):
        assert report["W%d" % ph] < TOL, report
        assert report["cs%d" % ph] < TOL, report
        assert int(got["best"][ph]) == ref.best[ph]
        assert report["loco%d" % ph] < TOL, report
    return ref, got


def test_config1_example_qt(example_dir):
    """BASELINE.json configs[0]: example.bed, 2 QT phenotypes, --bsize 100."""
    E = example_dir
    _compare(orc.Step1Options(bed=os.path.join(E, "example"), pheno_file=os.path.join(E, "phenotype.txt"),
                              covar_file=os.path.join(E, "covariates.txt"), bsize=100))


def test_example_3chr_loco(example_dir):
    E = example_dir
    ref, got = _compare(orc.Step1Options(bed=os.path.join(E, "example_3chr"), pheno_file=os.path.join(E, "phenotype.txt"),
                                         covar_file=os.path.join(E, "covariates.txt"), bsize=100))
    assert np.allclose(got["loco"][0][:3, 0], [-0.2066951705, -0.1673228499, 0.1541021496], atol=1e-8)


def test_remove_exclude_no_covariates(example_dir):
    """--remove / --exclude (ind_ignore path, ragged last block 94) and the intercept-only basis."""
    E = example_dir
    _compare(orc.Step1Options(bed=os.path.join(E, "example"), pheno_file=os.path.join(E, "phenotype.txt"),
                              remove=[os.path.join(E, "fid_iid_to_remove.txt")],
                              exclude=[os.path.join(E, "snplist_rm.txt")], bsize=100))


def test_missing_genotypes_and_phenotypes(tmp_path):
    """Mean imputation path (the reference's fixtures contain no missing call) +
    missing phenotypes (masks) + N not a multiple of 4 + odd block size."""
    N, M = 1203, 700
    g = synth_dosages(M, N, miss_rate=0.03, seed=21)
    chroms = np.repeat([1, 2, 5], [300, 250, 150])
    pre = str(tmp_path / "syn")
    write_plink(pre, g, chroms, P=3, ncov=2, seed=4, missing_pheno=0.05)
    _compare(orc.Step1Options(bed=pre, pheno_file=pre + ".pheno", covar_file=pre + ".covar", bsize=130))


@pytest.mark.parametrize("P", [4, 7, 14])
def test_many_phenotypes_matrix_core_predictions(tmp_path, P):
    """R0*P > 16 prediction rows select the fp64-MFMA level-0 prediction kernel (2, 3 and 4 row blocks, and two
    phenotype groups for P = 14): missing genotypes, missing phenotypes, ragged blocks, --remove-free file order."""
    N, M = 1100, 420
    g = synth_dosages(M, N, miss_rate=0.02, seed=31 + P)
    pre = str(tmp_path / "mp")
    write_plink(pre, g, np.repeat([1, 4, 9], [150, 150, 120]), P=P, ncov=2, seed=14, missing_pheno=0.04)
    _compare(orc.Step1Options(bed=pre, pheno_file=pre + ".pheno", covar_file=pre + ".covar", bsize=110))


def test_many_phenotypes_fp64_matrix_core_kernel_kept(tmp_path, monkeypatch):
    """RG_PRED_F64=1 keeps the fp64-MFMA prediction kernel (pred.hip) instead of the exact i8 digit route (pred_i8.hip)."""
    monkeypatch.setenv("RG_PRED_F64", "1")
    N, M = 1100, 420
    g = synth_dosages(M, N, miss_rate=0.02, seed=38)
    pre = str(tmp_path / "mp")
    write_plink(pre, g, np.repeat([1, 4, 9], [150, 150, 120]), P=7, ncov=2, seed=14, missing_pheno=0.04)
    _compare(orc.Step1Options(bed=pre, pheno_file=pre + ".pheno", covar_file=pre + ".covar", bsize=110))


def test_many_phenotypes_full_width_blocks(tmp_path):
    """The i8 digit route at the block width of the BASELINE configurations (1,000 SNPs: two 512-SNP register passes per
    position), with and without missing calls in a block, against the oracle and against the fp64 kernel."""
    N, M = 900, 1600
    g = synth_dosages(M, N, miss_rate=0.0, seed=77)
    g[1000:] = synth_dosages(600, N, miss_rate=0.03, seed=78)       # the second block has missing calls, the first has none
    pre = str(tmp_path / "fw")
    write_plink(pre, g, np.repeat([1, 2], [1000, 600]), P=6, ncov=2, seed=9, missing_pheno=0.03)
    opt = orc.Step1Options(bed=pre, pheno_file=pre + ".pheno", covar_file=pre + ".covar", bsize=1000)
    ref, got = _compare(opt)
    os.environ["RG_PRED_F64"] = "1"
    try:
        got64 = gpu_step1(opt)
    finally:
        del os.environ["RG_PRED_F64"]
    for ph in range(6):
        assert rel_err(got["W"][ph], got64["W"][ph]) < 1e-12


def test_ref_first_and_three_folds(tmp_path):
    N, M = 640, 256
    g = synth_dosages(M, N, miss_rate=0.01, seed=5)
    pre = str(tmp_path / "rf")
    write_plink(pre, g, np.repeat([1, 2], [128, 128]), P=1, ncov=1, seed=9)
    _compare(orc.Step1Options(bed=pre, pheno_file=pre + ".pheno", covar_file=pre + ".covar", bsize=64,
                              ref_first=True, cv_folds=3))


def test_low_variance_snp_is_reported(tmp_path):
    from regenie_amd.engine import RgError
    N, M = 256, 64
    g = synth_dosages(M, N, seed=8)
    g[10, :] = 1                                              # monomorphic -> sd 0 (Data.cpp:207-209)
    pre = str(tmp_path / "lv")
    write_plink(pre, g, np.ones(M, int), P=1, ncov=1, seed=2)
    opt = orc.Step1Options(bed=pre, pheno_file=pre + ".pheno", covar_file=pre + ".covar", bsize=64)
    with pytest.raises(ValueError):
        orc.run_step1(opt)
    with pytest.raises(RgError) as ei:
        gpu_step1(opt)
    assert ei.value.code == -3 and "low variance" in str(ei.value)


def test_block_batching_is_invariant(example_dir, monkeypatch):
    """Processing blocks 1-at-a-time or 8-at-a-time, on one, two or three pipelines, must give bit-identical W
    (deterministic kernels; level 0 always takes the group-wise Cholesky path)."""
    E = example_dir
    opt = orc.Step1Options(bed=os.path.join(E, "example_3chr"), pheno_file=os.path.join(E, "phenotype.txt"),
                           covar_file=os.path.join(E, "covariates.txt"), bsize=100)
    monkeypatch.setenv("RG_NBLK", "1")
    a = gpu_step1(opt)
    monkeypatch.setenv("RG_NBLK", "8")
    b = gpu_step1(opt)
    # one pipeline, and three pipelines with two blocks per batch (batches dealt round-robin over three HIP streams)
    monkeypatch.setenv("RG_PIPELINES", "1")
    c = gpu_step1(opt)
    monkeypatch.setenv("RG_PIPELINES", "3")
    monkeypatch.setenv("RG_NBLK", "2")
    d = gpu_step1(opt)
    for ph in range(2):
        for other in (b, c, d):
            assert np.array_equal(a["W"][ph], other["W"][ph])
            assert np.array_equal(a["loco"][ph], other["loco"][ph])


def test_level1_on_injected_predictors(example_dir):
    """File-seam analogue (--run-l1 on externally produced level-0 predictors): feed the
    oracle's W into the GPU level 1 and compare."""
    from regenie_amd.engine import Step1Engine, loco_from_predictions
    E = example_dir
    opt = orc.Step1Options(bed=os.path.join(E, "example_3chr"), pheno_file=os.path.join(E, "phenotype.txt"),
                           covar_file=os.path.join(E, "covariates.txt"), bsize=100)
    ref = orc.run_step1(opt)
    prep = ref.prep
    B, R0 = len(ref.blocks), 5
    eng = Step1Engine(0)
    eng.set_problem(X=prep.X, Y=prep.Y, mask=prep.mask, ind_in_analysis=prep.ind_in_analysis, cv_sizes=ref.cv_sizes,
                    lam=ref.lam, neff=prep.Neff, n_file=prep.n_file, n_blocks_total=B, max_block_size=100)
    for b in range(B):
        for ph in range(2):
            eng.set_w(b, ph, ref.W[ph][:, b * R0:(b + 1) * R0])
    chrcols = orc.chr_columns(ref.blocks, ref.chr_read, R0)
    cs, best, pred = eng.l1_qt(np.stack(ref.tau), [nn for (_, _, nn) in chrcols])
    for ph in range(2):
        assert rel_err(cs[ph], ref.cumsum[ph][:5]) < 1e-10
        loco = loco_from_predictions(pred[ph], [c for (c, _, _) in chrcols])
        assert rel_err(loco, ref.loco[ph]) < 1e-10
    eng.close()
