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
