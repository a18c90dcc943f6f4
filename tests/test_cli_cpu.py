"""Host-driver input handling that runs before any device is touched (so it is checkable without a GPU):
gzipped text inputs (Files.cpp:38-160) and the .pvar.gz / .psam.gz fallback (Geno.cpp:783, :952)."""
import gzip
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "regenie_amd", "bin", "regenie-amd")


@pytest.fixture(scope="module", autouse=True)
def _built():
    from regenie_amd import build
    build.build()


def _log_until_device(args, cwd):
    r = subprocess.run([BIN] + args, cwd=cwd, capture_output=True, text=True, timeout=120)
    out = r.stdout
    body = out[out.index("Fitting null model"):] if "Fitting null model" in out else out
    cut = body.find("ERROR: no MI355X")
    body = body[:cut] if cut >= 0 else body
    return r, "".join(ln for ln in body.splitlines(True) if "ms since start)" not in ln)     # wall-clock marks differ run to run


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


def test_gz_text_inputs_read_like_plain_ones(example_dir, tmp_path):
    if not _no_gpu():
        pytest.skip("GPU present: covered end to end by tests/test_cli_gpu.py::test_cli_gz_mode_of_the_reference_test")
    E = example_dir
    base = ["--step", "1", "--bed", os.path.join(E, "example"), "--bsize", "100", "--bt", "--out", str(tmp_path / "o")]
    r1, a = _log_until_device(base + ["--covarFile", os.path.join(E, "covariates.txt"), "--phenoFile", os.path.join(E, "phenotype_bin.txt")], str(tmp_path))
    # also: --remove list given gzipped
    with open(os.path.join(E, "fid_iid_to_remove.txt"), "rb") as fi, gzip.open(str(tmp_path / "rm.txt.gz"), "wb") as fo:
        fo.write(fi.read())
    r2, b = _log_until_device(base + ["--gz", "--covarFile", os.path.join(E, "covariates.txt.gz"), "--phenoFile",
                                      os.path.join(E, "phenotype_bin.txt.gz")], str(tmp_path))
    assert "n_pheno = 2" in a and "ERROR: no MI355X" in r1.stdout and "ERROR: no MI355X" in r2.stdout
    assert a.replace(".txt]", ".txt.gz]") == b   # same samples, same null logistic fits -- only the file names differ
    r3, c = _log_until_device(base + ["--covarFile", os.path.join(E, "covariates.txt.gz"), "--phenoFile", os.path.join(E, "phenotype_bin.txt.gz"),
                                      "--remove", str(tmp_path / "rm.txt.gz")], str(tmp_path))
    r4, d = _log_until_device(base + ["--covarFile", os.path.join(E, "covariates.txt.gz"), "--phenoFile", os.path.join(E, "phenotype_bin.txt.gz"),
                                      "--remove", os.path.join(E, "fid_iid_to_remove.txt")], str(tmp_path))
    assert "number of genotyped individuals remaining in the analysis = 494" in c and c == d


def test_pvar_psam_gz_fallback(example_dir, tmp_path):
    for ext in ("pgen",):
        shutil.copy(os.path.join(example_dir, "example." + ext), str(tmp_path / ("x." + ext)))
    for ext in ("pvar", "psam"):
        with open(os.path.join(example_dir, "example." + ext), "rb") as fi, gzip.open(str(tmp_path / ("x.%s.gz" % ext)), "wb") as fo:
            fo.write(fi.read())
    r, log = _log_until_device(["--step", "1", "--pgen", str(tmp_path / "x"), "--phenoFile", os.path.join(example_dir, "phenotype.txt"),
                                "--bsize", "100", "--out", str(tmp_path / "o")], str(tmp_path))
    assert "x.psam.gz] n_samples = 500" in log and "x.pvar.gz] n_snps = 1000" in log and "ERROR: incorrectly" not in r.stdout


def test_step2_usage_errors(example_dir, tmp_path):
    """--step 2: what is refused at the command line (before any file or device is touched) and how it is said."""
    E = example_dir
    base = ["--step", "2", "--bed", os.path.join(E, "example"), "--phenoFile", os.path.join(E, "phenotype_bin.txt"), "--bsize", "200", "--out", "x"]

    def err(extra):
        r = subprocess.run([BIN] + base + extra, cwd=str(tmp_path), capture_output=True, text=True, timeout=60)
        assert r.returncode != 0
        return r.stdout + r.stderr

    assert "option '--pred' is required" in err(["--bt"])
    assert "cannot use both" in err(["--bt", "--pred", "p.list", "--spa", "--firth", "--approx"])
    assert "applies to binary traits" in err(["--qt", "--pred", "p.list", "--firth", "--approx"])
    assert "minimum MAC must be at least 0.5" in err(["--qt", "--pred", "p.list", "--minMAC", "0.1"])


def test_t2e_options_and_phenotype_checks(example_dir, tmp_path):
    """--t2e: the option rules of the reference (Regenie.cpp:570-587, :1197-1201) and the checks of the (time, event) pairs
    (Pheno.cpp:262-283) happen before a device is needed; a well-formed run gets as far as the null Cox model and the device."""
    E = example_dir
    fam = [ln.split() for ln in open(os.path.join(E, "example_3chr.fam"))]

    def pheno(path, bad=None):
        with open(path, "w") as fh:
            fh.write("FID IID T1 E1\n")
            for i, t in enumerate(fam):
                tv, ev = "%g" % (1.0 + (i * 37 % 101) / 10.0), str(i % 3 == 0 and 1 or 0)
                if bad == "neg" and i == 5: tv = "-2"
                if bad == "censor" and i == 5: ev = "2"
                if bad == "missing_event" and i == 5: ev = "NA"
                if i == 7: tv, ev = "NA", "NA"
                fh.write("%s %s %s %s\n" % (t[0], t[1], tv, ev))

    base = ["--step", "1", "--bed", os.path.join(E, "example_3chr"), "--covarFile", os.path.join(E, "covariates.txt"), "--bsize", "100", "--out", "x"]

    def run(extra, bad=None):
        pheno(str(tmp_path / "t.txt"), bad)
        r = subprocess.run([BIN] + base + ["--phenoFile", str(tmp_path / "t.txt")] + extra, cwd=str(tmp_path), capture_output=True, text=True, timeout=60)
        assert r.returncode != 0
        return r.stdout + r.stderr

    assert "You must specify both '--phenoColList' and '--eventColList'" in run(["--t2e", "--phenoColList", "T1"])
    assert "must be used with '--t2e'" in run(["--qt", "--phenoColList", "T1", "--eventColList", "E1"])
    assert "You must specify TTE phenotypes using '--phenoColList'" in run(["--t2e", "--phenoCol", "T1", "--eventColList", "E1"])
    # the two undocumented level-1 switches are parsed (Regenie.cpp:366-367); without a GPU the run ends where every run does
    assert "no MI355X / HIP device" in run(["--t2e", "--phenoColList", "T1", "--eventColList", "E1", "--t2e-event-l0", "--t2e-l1-pi6"])
    ok = ["--t2e", "--phenoColList", "T1", "--eventColList", "E1"]
    assert "a phenotype time value is <0" in run(ok, "neg")
    assert "a phenotype censor value is invalid" in run(ok, "censor")
    assert "missing censor with non-missing time" in run(ok, "missing_event")
    out = run(ok)
    assert "n_pheno = 1" in out and "fitting null cox regression on time-to-event phenotypes...done" in out and "no MI355X / HIP device" in out


def test_step2_reads_inputs_then_needs_a_gpu(example_dir, tmp_path):
    """Step 2 on the .bgen example up to the device: the LOCO list and files are parsed (check_blup / blup_read), the variant table
    carries positions and alleles, and without a GPU the run ends with the library's error, not a CPU fallback."""
    E = example_dir
    R = os.path.join(ROOT, "tests", "golden", "ref_outputs", "qt_kfold_3chr")
    with open(str(tmp_path / "pred.list"), "w") as pl:
        for k in (1, 2):
            fn = str(tmp_path / ("ref_%d.loco" % k))
            open(fn, "wb").write(gzip.open(os.path.join(R, "out_%d.loco.gz" % k), "rb").read())
            pl.write("Y%d %s\n" % (k, fn))
    os.symlink(os.path.join(E, "example_3chr.bgen"), str(tmp_path / "ex3.bgen"))
    args = ["--step", "2", "--bgen", str(tmp_path / "ex3.bgen"), "--sample", os.path.join(E, "example_3chr.sample"), "--phenoFile", os.path.join(E, "phenotype.txt"),
            "--covarFile", os.path.join(E, "covariates.txt"), "--bsize", "200", "--qt", "--pred", str(tmp_path / "pred.list"), "--out", "s2"]
    r = subprocess.run([BIN] + args, cwd=str(tmp_path), capture_output=True, text=True, timeout=120)
    assert "n_snps" in r.stdout or "variants" in r.stdout
    if _no_gpu():
        assert r.returncode != 0 and "no MI355X / HIP device available" in r.stdout
        assert not os.path.exists(str(tmp_path / "s2_Y1.regenie")) or os.path.getsize(str(tmp_path / "s2_Y1.regenie")) == 0
    # a LOCO list that names an unknown file is an error before the device is needed
    open(str(tmp_path / "bad.list"), "w").write("Y1 %s\nY2 %s\n" % (str(tmp_path / "nope.loco"), str(tmp_path / "ref_2.loco")))
    r = subprocess.run([BIN] + args[:-4] + ["--pred", str(tmp_path / "bad.list"), "--out", "s3"], cwd=str(tmp_path), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "nope.loco" in r.stdout


def test_user_ridge_grids_are_checked_like_the_reference(example_dir, tmp_path):
    """--setl0 / --setl1 (get_unit_params, Regenie.cpp:1477-1495): values outside (0, 1) stop the run with regenie's message before any device is touched.
    (Sorted and de-duplicated as well: a drawn case with `--setl0 0.2334,0.2334,...` gave other numbers than regenie until round 5,
    tests/golden/fuzz_log.md; that part needs a GPU to show.)"""
    E = example_dir
    for opt, name in (("--setl0", "--l0"), ("--setl1", "--l1")):
        r = subprocess.run([BIN, "--step", "1", "--bed", os.path.join(E, "example"), "--phenoFile", os.path.join(E, "phenotype.txt"), "--covarFile",
                            os.path.join(E, "covariates.txt"), "--bsize", "100", opt, "0.5,1.5", "--out", str(tmp_path / "o")], capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and ("ERROR: must specify values for %s in (0,1)." % name) in r.stdout + r.stderr
