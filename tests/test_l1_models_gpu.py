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


def test_bt_kfold_fifty_phenotypes(tmp_path):
    """BASELINE configs[3]'s phenotype count: 50 binary traits (prevalence 30 %, 2 % missing values each) through level 0 -- 250 rows
    of (phenotype, ridge value) per block: five groups of the exact digit-plane prediction kernel -- and the K-fold logistic ridge of every
    trait, against the oracle: CV sums, selected ridge value, convergence flags and LOCO predictors of all 50."""
    N, M, P = 5200, 400, 50
    g = synth_dosages(M, N, miss_rate=0.005, seed=501)
    pre = str(tmp_path / "bt50")
    write_plink(pre, g, np.repeat([1, 2], [250, 150]), P=P, ncov=2, seed=21, missing_pheno=0.02, binary=True)
    opt = orc.Step1Options(bed=pre, pheno_file=pre + ".pheno", covar_file=pre + ".covar", bsize=100, bt=True)
    ref = orc.run_step1(opt)
    assert not ref.use_loocv
    got = gpu_step1_any(opt)
    assert not got["use_loocv"]
    _compare(ref, got, P, TOL_BT, 4)


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


# ---- count traits (--ct): Poisson ridge level 1, Step1_Models.cpp:1429-1758 ----------------------------------------------
def _count_pheno_file(example_dir, path, with_na_rows=True, seed=5):
    """Two count phenotypes on the example's samples: one driven by the example's first QT, one pure noise.  A row is
    either complete or missing both (a partially missing row would hit the reference's rate-with-missing-code quirk,
    oracle.tau_count)."""
    rng = np.random.default_rng(seed)
    lines = open(os.path.join(example_dir, "phenotype.txt")).read().split("\n")
    with open(path, "w") as fh:
        fh.write("FID IID C1 C2\n")
        for i, ln in enumerate(lines[1:]):
            t = ln.split()
            if not t:
                continue
            c1 = rng.poisson(np.exp(0.5 + 0.4 * float(t[2])))
            c2 = rng.poisson(2.0)
            if with_na_rows and i % 50 == 7:
                fh.write("%s %s NA NA\n" % (t[0], t[1]))
            else:
                fh.write("%s %s %d %d\n" % (t[0], t[1], c1, c2))


@pytest.mark.parametrize("loocv", [False, True])
def test_ct_poisson_example(example_dir, tmp_path, loocv):
    E = example_dir
    ph = str(tmp_path / "ct.txt")
    _count_pheno_file(E, ph)
    opt = orc.Step1Options(bed=os.path.join(E, "example"), pheno_file=ph, covar_file=os.path.join(E, "covariates.txt"),
                           remove=[os.path.join(E, "fid_iid_to_remove.txt")], bsize=100, ct=True, loocv=loocv)
    ref = orc.run_step1(opt)
    got = gpu_step1_any(opt)
    assert got["use_loocv"] == loocv and ref.use_loocv == loocv
    assert all(ref.converged)
    _compare(ref, got, 2, TOL_BT, 6)
    # the log label of a count trait inverts the tau map back to the h grid (Data.cpp:1039-1054)
    rate = got["prep"].Y_raw[:, 0].sum() / got["prep"].Neff[0]
    lines = orc.cv_table(np.asarray(got["cumsum"][0]), got["prep"].Neff[0], got["L"], got["tau"][0], False, int(got["best"][0]), ct_rate=rate)
    assert [ln.split(":")[0].strip() for ln in lines] == ["0.01", "0.25", "0.5", "0.75", "0.99"]
    assert all("MSE" not in ln and "-logLik/N" in ln for ln in lines)


def test_ct_poisson_loco_on_device_and_ragged(tmp_path):
    N, M = 640, 300
    g = synth_dosages(M, N, miss_rate=0.01, seed=13)
    pre = str(tmp_path / "ct")
    write_plink(pre, g, np.repeat([1, 3, 4], [110, 100, 90]), P=2, ncov=2, seed=4)
    rng = np.random.default_rng(2)
    rows = open(pre + ".pheno").read().split("\n")
    with open(pre + ".ct", "w") as fh:
        fh.write("FID IID K1\n")
        for ln in rows[1:]:
            t = ln.split()
            if t:
                fh.write("%s %s %d\n" % (t[0], t[1], rng.poisson(np.exp(0.2 + 0.5 * float(t[2]))) if t[2] != "NA" else 0))
    opt = orc.Step1Options(bed=pre, pheno_file=pre + ".ct", covar_file=pre + ".covar", bsize=70, ct=True)
    ref = orc.run_step1(opt)
    got = gpu_step1_any(opt, loco_on_device=True)
    _compare(ref, got, 1, TOL_BT, 6)
