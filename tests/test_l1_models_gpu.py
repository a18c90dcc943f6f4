"""Level-1 models beyond QT K-fold, through the C ABI against the oracle: QT LOOCV, BT (logistic ridge)
LOOCV and BT K-fold.  The BT LOOCV case is the reference's own Step-1 test command
(test/test_bash.sh:62-89), whose log must carry `0.4504` on the `min value` line."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
pytest.importorskip("torch")

from oracle import regenie_step1 as orc  # noqa: E402
from tests.util import gpu_step1_any, oracle_step1_any, rel_err, synth_dosages, write_plink  # noqa: E402

TOL = 1e-8      # BASELINE.json asks 1e-5 on the LOCO predictors
TOL_BT = 1e-6   # iterative logistic fits stop on max|score| < 1e-4: the iterates agree to ~1e-9


def _compare(ref, got, P, tol, nrow):
    for ph in range(P):
        assert bool(got["converged"][ph]) == bool(ref.converged[ph])
        cs_ref = np.asarray(ref.cumsum[ph])[:nrow]
        cs_got = np.asarray(got["cumsum"][ph])[:nrow]
        assert np.max(np.abs(cs_got - cs_ref) / (1.0 + np.abs(cs_ref))) < tol, (ph, cs_got, cs_ref)
        assert int(got["best"][ph]) == ref.best[ph]
        assert rel_err(got["loco"][ph], ref.loco[ph]) < tol


def test_qt_loocv_example_3chr(example_dir):
    E = example_dir
    opt = orc.Step1Options(bed=os.path.join(E, "example_3chr"), pheno_file=os.path.join(E, "phenotype.txt"),
                           covar_file=os.path.join(E, "covariates.txt"), bsize=100, loocv=True)
    ref = orc.run_step1(opt)
    got = gpu_step1_any(opt)
    assert got["use_loocv"] and ref.use_loocv
    _compare(ref, got, 2, TOL, 5)


def test_qt_loocv_missing_ragged(tmp_path):
    N, M = 700, 330
    g = synth_dosages(M, N, miss_rate=0.02, seed=41)
    pre = str(tmp_path / "ql")
    write_plink(pre, g, np.repeat([1, 2, 5], [130, 100, 100]), P=2, ncov=2, seed=8, missing_pheno=0.05)
    opt = orc.Step1Options(bed=pre, pheno_file=pre + ".pheno", covar_file=pre + ".covar", bsize=70, loocv=True)
    ref = orc.run_step1(opt)
    got = gpu_step1_any(opt)
    _compare(ref, got, 2, TOL, 5)


def test_bt_loocv_reference_command(example_dir):
    E = example_dir
    opt = orc.Step1Options(bed=os.path.join(E, "example"), pheno_file=os.path.join(E, "phenotype_bin.txt"),
                           covar_file=os.path.join(E, "covariates.txt"), remove=[os.path.join(E, "fid_iid_to_remove.txt")],
                           exclude=[os.path.join(E, "snplist_rm.txt")], bsize=100, bt=True)
    ref = orc.run_step1(opt)
    got = gpu_step1_any(opt)
    assert got["use_loocv"] and ref.use_loocv
    _compare(ref, got, 2, TOL_BT, 6)
    # the reference's known answer, from the GPU sums (Data.cpp:1042-1077 log line)
    lines = orc.cv_table(np.asarray(got["cumsum"][1]), got["prep"].Neff[1], got["L"], got["tau"][1], True,
                         int(got["best"][1]))
    assert any(("0.4504" in ln and "min value" in ln) for ln in lines), lines


def test_bt_kfold_example_3chr(example_dir):
    E = example_dir
    opt = orc.Step1Options(bed=os.path.join(E, "example_3chr"), pheno_file=os.path.join(E, "phenotype_bin.txt"),
                           covar_file=os.path.join(E, "covariates.txt"), bsize=100, bt=True)
    ref = oracle_step1_any(opt, force_kfold=True)
    got = gpu_step1_any(opt, force_kfold=True)
    assert not got["use_loocv"]
    _compare(ref, got, 2, TOL_BT, 6)


def test_bt_kfold_missing_pheno(example_dir):
    E = example_dir
    opt = orc.Step1Options(bed=os.path.join(E, "example_3chr"), pheno_file=os.path.join(E, "phenotype_bin_wNA.txt"),
                           covar_file=os.path.join(E, "covariates.txt"), bsize=200, bt=True, cv_folds=3)
    ref = oracle_step1_any(opt, force_kfold=True)
    got = gpu_step1_any(opt, force_kfold=True)
    _compare(ref, got, 2, TOL_BT, 6)


def test_bt_kfold_native_above_5000(tmp_path):
    """--bt with more than 5,000 samples keeps K-fold CV by the reference's own rule (Data.cpp:353): the logistic
    ridge IRLS with all K fold models in lock-step launches, on synthetic 0/1 traits with missing values."""
    N, M = 5300, 600
    g = synth_dosages(M, N, miss_rate=0.01, seed=77)
    pre = str(tmp_path / "btk")
    write_plink(pre, g, np.repeat([1, 2, 7], [250, 200, 150]), P=2, ncov=2, seed=12, missing_pheno=0.03, binary=True)
    opt = orc.Step1Options(bed=pre, pheno_file=pre + ".pheno", covar_file=pre + ".covar", bsize=100, bt=True)
    ref = orc.run_step1(opt)
    assert not ref.use_loocv
    got = gpu_step1_any(opt)
    assert not got["use_loocv"]
    _compare(ref, got, 2, TOL_BT, 6)


@pytest.mark.parametrize("kind", ["qt_kfold", "qt_loocv", "bt_loocv"])
def test_loco_output_mode_matches_host_assembly(example_dir, kind):
    """rg_set_loco_output: the device-side LOCO assembly (write_predictions, Data.cpp:1846-1858) is bit-identical to the
    host-side assembly of the per-chromosome predictions, for every level-1 entry point."""
    E = example_dir
    opt = orc.Step1Options(bed=os.path.join(E, "example" if kind == "bt_loocv" else "example_3chr"),
                           pheno_file=os.path.join(E, "phenotype_bin.txt" if kind == "bt_loocv" else "phenotype.txt"),
                           covar_file=os.path.join(E, "covariates.txt"), bsize=100, bt=(kind == "bt_loocv"),
                           loocv=(kind == "qt_loocv"))
    ref = gpu_step1_any(opt)
    got = gpu_step1_any(opt, loco_on_device=True)
    for ph in range(len(ref["loco"])):
        assert got["loco"][ph].shape == ref["loco"][ph].shape
        assert np.array_equal(ref["loco"][ph], got["loco"][ph])
