"""Drop-in check of the C++ host driver (GPU): `regenie-amd --step 1 ...` must produce the reference's
output files (<out>_pred.list, <out>_<k>.loco, <out>.log) in the reference's format (src/Data.cpp:1926-1975)
with values equal to the oracle's to the 6 significant digits the format carries."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
pytest.importorskip("torch")

from oracle import regenie_step1 as orc  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "regenie_amd", "bin", "regenie-amd")


def _parse_loco(path):
    lines = open(path).read().split("\n")
    assert lines[-1] == ""
    hdr = lines[0].split(" ")
    assert hdr[0] == "FID_IID" and hdr[-1] == ""
    rows = []
    for ln in lines[1:-1]:
        t = ln.split(" ")
        assert t[-1] == ""
        rows.append([np.nan if v == "NA" else float(v) for v in t[1:-1]])
    return hdr[1:-1], [ln.split(" ")[0] for ln in lines[1:-1]], np.array(rows), lines


def _run(args, cwd):
    r = subprocess.run([BIN] + args, cwd=cwd, capture_output=True, text=True, timeout=300)
    return r


def test_cli_matches_oracle_files(example_dir, tmp_path):
    E = example_dir
    args = ["--step", "1", "--bed", os.path.join(E, "example_3chr"), "--phenoFile", os.path.join(E, "phenotype.txt"),
            "--covarFile", os.path.join(E, "covariates.txt"), "--bsize", "100", "--lowmem", "--lowmem-prefix", "tmp_rg",
            "--out", str(tmp_path / "gpu")]
    r = _run(args, str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    opt = orc.Step1Options(bed=os.path.join(E, "example_3chr"), pheno_file=os.path.join(E, "phenotype.txt"),
                           covar_file=os.path.join(E, "covariates.txt"), bsize=100, out=str(tmp_path / "ref"))
    ref = orc.run_step1(opt, write_files=True)
    for k in (1, 2):
        ids_g, chr_g, val_g, lines_g = _parse_loco(str(tmp_path / ("gpu_%d.loco" % k)))
        ids_r, chr_r, val_r, lines_r = _parse_loco(str(tmp_path / ("ref_%d.loco" % k)))
        assert ids_g == ids_r and chr_g == chr_r == [str(c) for c in range(1, 24)]
        assert val_g.shape == val_r.shape == (23, 500)
        assert np.nanmax(np.abs(val_g - val_r)) <= 2e-6 * np.nanmax(np.abs(val_r))   # 6 printed digits
        same = sum(a == b for a, b in zip(lines_g, lines_r))
        assert same >= 20, "text of most chromosome rows should be byte-identical (%d/25)" % same
    pl = open(str(tmp_path / "gpu_pred.list")).read().split("\n")
    assert pl[0] == "Y1 " + str(tmp_path / "gpu_1.loco") and pl[1] == "Y2 " + str(tmp_path / "gpu_2.loco")
    log = open(str(tmp_path / "gpu.log")).read()
    assert log.count("<- min value") == 2
    for ph in range(2):
        j = ref.best[ph]
        cs = ref.cumsum[ph]
        mse = (cs[2, j] + cs[3, j] - 2 * cs[4, j]) / ref.prep.Neff[ph]
        assert ("MSE = %s<- min value" % orc.cpp_double(mse)) in log


@pytest.mark.parametrize("stage", ["1", "2", "given_up"])
def test_cli_bed_staged_in_device_memory_equals_streamed(example_dir, tmp_path, stage):
    """The whole .bed copied to the device up front (BedStage of host/driver_step1.cpp over rg_stage_alloc / rg_stage_copy; the default for
    files of several GB, forced here: RG_INGEST_STAGE=1 the pread threads + page-locked ring, =2 pageable copies from the mapping) and level
    0 reading the rows in place must give the files of the batch-by-batch ingest byte for byte."""
    E = example_dir
    common = ["--step", "1", "--bed", os.path.join(E, "example_3chr"), "--phenoFile", os.path.join(E, "phenotype.txt"),
              "--covarFile", os.path.join(E, "covariates.txt"), "--bsize", "100"]
    outs = {}
    # "given_up": the stage is started and then cancelled before level 0 (what the driver does when W and the workspaces would not fit beside it)
    staged_env = {"RG_INGEST_STAGE": "1", "RG_STAGE_FORCE_CANCEL": "1"} if stage == "given_up" else {"RG_INGEST_STAGE": stage}
    for name, env in (("streamed", {"RG_INGEST_STAGE": "0"}), ("staged", staged_env)):
        r = subprocess.run([BIN] + common + ["--out", str(tmp_path / name)], cwd=str(tmp_path), capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, RG_TEARDOWN="1", **env))
        assert r.returncode == 0, r.stdout + r.stderr
        outs[name] = r.stdout
    if stage == "given_up":
        assert "is not kept in device memory" in outs["staged"] and "copied to the GPU from the mapped file" in outs["staged"]
    else:
        assert "read from the copy of the file in device memory" in outs["staged"]
    assert "copied to the GPU from the mapped file" in outs["streamed"]
    for k in (1, 2):
        assert open(str(tmp_path / ("staged_%d.loco" % k)), "rb").read() == open(str(tmp_path / ("streamed_%d.loco" % k)), "rb").read()


def test_cli_embedded_right_hand_sides_equal_separate_rows(example_dir, tmp_path):
    """Level 0 keeps the right-hand sides of a block's ridge systems in the padding rows of the systems' last tile whenever they fit
    (regenie_amd/csrc/chol.hip, "embedded right-hand sides"); RG_NO_EMBED=1 keeps them in a tile row of their own.  Same predictions to
    the printed digits (the forward substitution of the last tile columns runs in another kernel: rounding differs)."""
    E = example_dir
    outs = {}
    for name, env in (("embedded", {}), ("separate", {"RG_NO_EMBED": "1"})):
        args = ["--step", "1", "--bed", os.path.join(E, "example_3chr"), "--phenoFile", os.path.join(E, "phenotype.txt"),
                "--covarFile", os.path.join(E, "covariates.txt"), "--bsize", "100", "--out", str(tmp_path / name)]
        r = subprocess.run([BIN] + args, cwd=str(tmp_path), capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stdout + r.stderr
        outs[name] = [_parse_loco(str(tmp_path / ("%s_%d.loco" % (name, k))))[2] for k in (1, 2)]
    for a, b in zip(outs["embedded"], outs["separate"]):
        assert np.nanmax(np.abs(a - b)) <= 2e-6 * np.nanmax(np.abs(b))


def test_cli_remove_exclude_prs(example_dir, tmp_path):
    E = example_dir
    args = ["--step", "1", "--bed", os.path.join(E, "example"), "--phenoFile", os.path.join(E, "phenotype.txt"),
            "--remove", os.path.join(E, "fid_iid_to_remove.txt"), "--exclude", os.path.join(E, "snplist_rm.txt"),
            "--bsize", "100", "--print-prs", "--use-relative-path", "--phenoCol", "Y2", "--out", "o"]
    r = _run(args, str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    opt = orc.Step1Options(bed=os.path.join(E, "example"), pheno_file=os.path.join(E, "phenotype.txt"),
                           remove=[os.path.join(E, "fid_iid_to_remove.txt")], exclude=[os.path.join(E, "snplist_rm.txt")],
                           bsize=100, pheno_cols=["Y2"])
    ref = orc.run_step1(opt)
    ids, chrs, val, _ = _parse_loco(str(tmp_path / "o_1.loco"))
    assert len(ids) == 494
    order = sorted(range(494), key=lambda i: ref.prep.ids[i])
    exp = ref.loco[0][order].T
    assert np.max(np.abs(val - exp)) <= 2e-6 * np.max(np.abs(exp))
    assert open(str(tmp_path / "o_pred.list")).read() == "Y2 o_1.loco\n"
    assert os.path.exists(str(tmp_path / "o_1.prs")) and open(str(tmp_path / "o_prs.list")).read() == "Y2 o_1.prs\n"


def test_cli_errors_like_reference(example_dir, tmp_path):
    E = example_dir
    r = _run(["--step", "1", "--bed", os.path.join(E, "example"), "--phenoFile", os.path.join(E, "phenotype.txt")], str(tmp_path))
    assert r.returncode != 0 and "ERROR: must specify the block size using '--bsize'." in r.stdout
    r = _run(["--step", "1", "--bed", os.path.join(E, "nope"), "--phenoFile", os.path.join(E, "phenotype.txt"), "--bsize", "10"], str(tmp_path))
    assert r.returncode != 0 and "ERROR:" in r.stdout


def test_cli_bt_reference_command(example_dir, tmp_path):
    """The reference's own Step-1 test (test/test_bash.sh:62-89): --bt on 494 samples (automatic LOOCV),
    --lowmem; its log must show 0.4504 on the `min value` line of Y2."""
    E = example_dir
    args = ["--step", "1", "--bed", os.path.join(E, "example"), "--exclude", os.path.join(E, "snplist_rm.txt"),
            "--covarFile", os.path.join(E, "covariates.txt"), "--phenoFile", os.path.join(E, "phenotype_bin.txt"),
            "--remove", os.path.join(E, "fid_iid_to_remove.txt"), "--bsize", "100", "--bt", "--lowmem",
            "--lowmem-prefix", "tmp_rg", "--out", str(tmp_path / "gpu")]
    r = _run(args, str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    log = open(str(tmp_path / "gpu.log")).read()
    assert any(("0.4504" in ln and "min value" in ln) for ln in log.split("\n")), log
    assert "using LOOCV instead of 5-fold CV" in log
    opt = orc.Step1Options(bed=os.path.join(E, "example"), pheno_file=os.path.join(E, "phenotype_bin.txt"),
                           covar_file=os.path.join(E, "covariates.txt"), remove=[os.path.join(E, "fid_iid_to_remove.txt")],
                           exclude=[os.path.join(E, "snplist_rm.txt")], bsize=100, bt=True, out=str(tmp_path / "ref"))
    ref = orc.run_step1(opt, write_files=True)
    for ln in ref.log:   # every per-tau line of the oracle's log appears verbatim in the driver's log
        if "Rsq" in ln:
            assert ln in log, ln
    for k in (1, 2):
        ids_g, chr_g, val_g, lines_g = _parse_loco(str(tmp_path / ("gpu_%d.loco" % k)))
        ids_r, chr_r, val_r, lines_r = _parse_loco(str(tmp_path / ("ref_%d.loco" % k)))
        assert ids_g == ids_r and chr_g == chr_r
        assert np.nanmax(np.abs(val_g - val_r)) <= 2e-6 * np.nanmax(np.abs(val_r))


def test_cli_qt_loocv(example_dir, tmp_path):
    E = example_dir
    args = ["--step", "1", "--bed", os.path.join(E, "example_3chr"), "--phenoFile", os.path.join(E, "phenotype.txt"),
            "--covarFile", os.path.join(E, "covariates.txt"), "--bsize", "100", "--loocv", "--out", str(tmp_path / "gpu")]
    r = _run(args, str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    opt = orc.Step1Options(bed=os.path.join(E, "example_3chr"), pheno_file=os.path.join(E, "phenotype.txt"),
                           covar_file=os.path.join(E, "covariates.txt"), bsize=100, loocv=True, out=str(tmp_path / "ref"))
    ref = orc.run_step1(opt, write_files=True)
    log = open(str(tmp_path / "gpu.log")).read()
    for ln in ref.log:
        if "Rsq" in ln:
            assert ln in log, ln
    for k in (1, 2):
        _, _, val_g, _ = _parse_loco(str(tmp_path / ("gpu_%d.loco" % k)))
        _, _, val_r, _ = _parse_loco(str(tmp_path / ("ref_%d.loco" % k)))
        assert np.nanmax(np.abs(val_g - val_r)) <= 2e-6 * np.nanmax(np.abs(val_r))


def test_cli_split_l0_equals_single_run(example_dir, tmp_path):
    """The reference's self-consistency test (test/test_bash.sh:91-138) on the C++ driver: --split-l0 into 4 jobs,
    4 x --run-l0 (raw double level-0 files, global M for lambda), --run-l1; both .loco files must be byte-identical
    to the single-process run of the same (BT, automatic LOOCV) command."""
    E = example_dir
    base = ["--step", "1", "--bed", os.path.join(E, "example"), "--exclude", os.path.join(E, "snplist_rm.txt"),
            "--covarFile", os.path.join(E, "covariates.txt"), "--phenoFile", os.path.join(E, "phenotype_bin.txt"),
            "--remove", os.path.join(E, "fid_iid_to_remove.txt"), "--bsize", "100", "--bt", "--lowmem",
            "--lowmem-prefix", "tmp_rg"]
    r = _run(base + ["--out", str(tmp_path / "fit_bin_out")], str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    pre = str(tmp_path / "fit_bin_parallel")
    r = _run(base + ["--split-l0", pre + ",4", "--out", str(tmp_path / "fit_bin_l0")], str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    master = open(pre + ".master").read().split("\n")
    assert master[0] == "994 100"                                 # <n_variants> <bsize> (1000 - 6 excluded)
    assert [ln.split(" ")[1:] for ln in master[1:5]] == [["3", "300"], ["3", "300"], ["2", "200"], ["2", "194"]]
    assert len(open(pre + "_job4.snplist").read().split()) == 194
    for job in range(1, 5):
        r = _run(base + ["--run-l0", "%s.master,%d" % (pre, job), "--out", str(tmp_path / "fit_bin_l0")], str(tmp_path))
        assert r.returncode == 0, r.stdout + r.stderr
        nb = int(master[job].split(" ")[1])
        assert os.path.getsize("%s_job%d_l0_Y1" % (pre, job)) == 8 * 494 * nb * 5   # raw double N x (blocks*R0)
    r = _run(base + ["--run-l1", pre + ".master", "--out", str(tmp_path / "fit_bin_l1")], str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    for k in (1, 2):
        a = open(str(tmp_path / ("fit_bin_out_%d.loco" % k)), "rb").read()
        b = open(str(tmp_path / ("fit_bin_l1_%d.loco" % k)), "rb").read()
        assert a == b, "split-l0 run differs from the single run for phenotype %d" % k
    assert not os.path.exists(pre + "_job1_l0_Y1")                # removed after --run-l1 (no --keep-l0)


def test_cli_pgen_equals_bed(example_dir, tmp_path):
    """`--pgen example` and `--bed example` are the same genotypes (the reference's own test test/test_bash.sh:411-433
    relies on that pair): every output file of the two runs must be byte-identical, with sample and variant filters."""
    E = example_dir
    common = ["--step", "1", "--exclude", os.path.join(E, "snplist_rm.txt"), "--covarFile", os.path.join(E, "covariates.txt"),
              "--phenoFile", os.path.join(E, "phenotype.txt"), "--remove", os.path.join(E, "fid_iid_to_remove.txt"),
              "--bsize", "100"]
    r = _run(common + ["--bed", os.path.join(E, "example"), "--out", str(tmp_path / "b")], str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    r = _run(common + ["--pgen", os.path.join(E, "example"), "--out", str(tmp_path / "p")], str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    assert " * pgen" in r.stdout and "n_snps = 1000" in r.stdout
    for k in (1, 2):
        a = open(str(tmp_path / ("b_%d.loco" % k)), "rb").read()
        b = open(str(tmp_path / ("p_%d.loco" % k)), "rb").read()
        assert len(a) > 1000 and a == b


def test_cli_pgen_compressed_records(tmp_path):
    """A .pgen that uses every record type (LD-compressed, one-bit, difflists ...) against the same genotypes as .bed."""
    from oracle import pgen as opg
    from tests.test_pgen import synth
    m, n = 400, 600
    g, vts = synth(m, n, seed=77)
    rng = np.random.default_rng(4)
    # keep the variants polymorphic enough for the level-0 blocks: replace monomorphic rows by common 2-bit ones
    for j in range(m):
        gj = g[j][g[j] != 3]
        if gj.size < n // 2 or gj.std() < 0.2:
            g[j] = (rng.random(n) < 0.3).astype(np.uint8) + (rng.random(n) < 0.3).astype(np.uint8)
            vts[j] = 0
    # re-derive LD-compressed records after the edits (their base may have changed)
    prev = None
    for j in range(m):
        if (vts[j] & 6) == 2:
            d = (g[j] if vts[j] == 2 else opg._invert(g[j])) != prev
            if prev is None or d.sum() > n // 8:
                vts[j] = 0
        if (vts[j] & 6) != 2:
            prev = g[j]
    assert {1, 2, 3} <= set(vts)
    pfx = str(tmp_path / "syn")
    opg.write_pgen(pfx + ".pgen", g, vts)
    chroms = [1 + (j * 4) // m for j in range(m)]
    opg.write_pvar_psam(pfx, chroms, n)
    with open(pfx + ".bed", "wb") as fh:
        fh.write(b"\x6c\x1b\x01")
        for j in range(m):
            fh.write(opg._pack2(opg.PGEN_TO_BED[g[j]]))
    with open(pfx + ".bim", "w") as fh:
        for j in range(m):
            fh.write("%d\tv%d\t0\t%d\tC\tA\n" % (chroms[j], j + 1, j + 1))
    with open(pfx + ".fam", "w") as fh:
        for i in range(n):
            fh.write("%d %d 0 0 0 -9\n" % (i + 1, i + 1))
    with open(pfx + ".pheno", "w") as fh:
        fh.write("FID IID Y1 Y2\n")
        y = rng.standard_normal((n, 2)) + 0.3 * (g[:40] % 3).sum(0)[:, None]
        for i in range(n):
            fh.write("%d %d %.6f %.6f\n" % (i + 1, i + 1, y[i, 0], y[i, 1]))
    common = ["--step", "1", "--phenoFile", pfx + ".pheno", "--bsize", "50"]
    r = _run(common + ["--bed", pfx, "--out", str(tmp_path / "b")], str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    r = _run(common + ["--pgen", pfx, "--out", str(tmp_path / "p")], str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    for k in (1, 2):
        assert open(str(tmp_path / ("b_%d.loco" % k)), "rb").read() == open(str(tmp_path / ("p_%d.loco" % k)), "rb").read()


def test_cli_gz_mode_of_the_reference_test(example_dir, tmp_path):
    """test/test_bash.sh run with Boost Iostreams (`fsuf=.gz`, `arg_gz=--gz`, :45-53, :64-76): the covariate and phenotype files
    are read gzipped and the .loco files are written gzipped.  Decompressed they must be the plain run's bytes, and
    _pred.list must point at the .gz files."""
    import gzip
    E = example_dir
    base = ["--step", "1", "--bed", os.path.join(E, "example"), "--exclude", os.path.join(E, "snplist_rm.txt"),
            "--remove", os.path.join(E, "fid_iid_to_remove.txt"), "--bsize", "100", "--bt", "--lowmem", "--lowmem-prefix", "tmp_rg"]
    plain = base + ["--covarFile", os.path.join(E, "covariates.txt"), "--phenoFile", os.path.join(E, "phenotype_bin.txt")]
    gz = base + ["--covarFile", os.path.join(E, "covariates.txt.gz"), "--phenoFile", os.path.join(E, "phenotype_bin.txt.gz"), "--gz"]
    r = _run(plain + ["--out", str(tmp_path / "a")], str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    r = _run(gz + ["--out", str(tmp_path / "z")], str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    for k in (1, 2):
        assert not os.path.exists(str(tmp_path / ("z_%d.loco" % k)))
        a = open(str(tmp_path / ("a_%d.loco" % k)), "rb").read()
        z = gzip.open(str(tmp_path / ("z_%d.loco.gz" % k)), "rb").read()
        assert len(a) > 1000 and a == z
    lines = open(str(tmp_path / "z_pred.list")).read().split("\n")
    assert lines[0] == "Y1 " + str(tmp_path / "z_1.loco.gz") and lines[1] == "Y2 " + str(tmp_path / "z_2.loco.gz")
    # a file named .gz that is not gzip data is read as plain text (Files::isGzipped checks the magic bytes)
    import shutil
    shutil.copy(os.path.join(E, "phenotype_bin.txt"), str(tmp_path / "fake.txt.gz"))
    r = _run(base + ["--covarFile", os.path.join(E, "covariates.txt"), "--phenoFile", str(tmp_path / "fake.txt.gz"),
                     "--out", str(tmp_path / "f")], str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    assert open(str(tmp_path / "f_1.loco"), "rb").read() == open(str(tmp_path / "a_1.loco"), "rb").read()


@pytest.mark.parametrize("loocv", [False, True])
def test_cli_ct_count_traits(example_dir, tmp_path, loocv):
    """`--ct`: count phenotypes, null Poisson offsets, Poisson ridge level 1 (Step1_Models.cpp:225-345, :1429-1758) -- the
    driver's files against the oracle's."""
    from tests.test_l1_models_gpu import _count_pheno_file
    E = example_dir
    ph = str(tmp_path / "ct.txt")
    _count_pheno_file(E, ph)
    args = ["--step", "1", "--bed", os.path.join(E, "example"), "--phenoFile", ph, "--covarFile", os.path.join(E, "covariates.txt"),
            "--bsize", "100", "--ct", "--out", str(tmp_path / "c")] + (["--loocv"] if loocv else [])
    r = _run(args, str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    opt = orc.Step1Options(bed=os.path.join(E, "example"), pheno_file=ph, covar_file=os.path.join(E, "covariates.txt"), bsize=100,
                           ct=True, loocv=loocv, out=str(tmp_path / "o"))
    ref = orc.run_step1(opt, write_files=True)
    assert "fitting null poisson regression" in r.stdout and "Level 1 ridge with poisson regression" in r.stdout
    for k in (1, 2):
        h1, ids1, v1, _ = _parse_loco(str(tmp_path / ("c_%d.loco" % k)))
        h2, ids2, v2, _ = _parse_loco(str(tmp_path / ("o_%d.loco" % k)))
        assert h1 == h2 and ids1 == ids2
        assert np.allclose(v1, v2, rtol=2e-5, atol=1e-7, equal_nan=True)
    # the per-tau table (label column inverts tau back to the h grid; no MSE column for counts)
    tab = [ln for ln in r.stdout.split("\n") if " : Rsq = " in ln]
    want = [ln for ln in ref.log if " : Rsq = " in ln]
    assert len(tab) == 10 and all("MSE" not in ln for ln in tab)
    for a, b in zip(tab, want):
        assert a.split(":")[0].strip() == b.split(":")[0].strip() and ("min value" in a) == ("min value" in b)
    # negative counts are refused with the reference's message (Pheno.cpp:317-318)
    bad = open(ph).read().replace("\n", "\n", 1).split("\n")
    t = bad[3].split()
    bad[3] = "%s %s -2 1" % (t[0], t[1])
    open(str(tmp_path / "bad.txt"), "w").write("\n".join(bad))
    r = _run(args[:5] + [str(tmp_path / "bad.txt")] + args[6:], str(tmp_path))
    assert r.returncode != 0 and "a phenotype value is <0 for individual: FID=%s IID=%s Y=-2" % (t[0], t[1]) in r.stdout


def test_cli_option_mix(example_dir, tmp_path):
    """Less-travelled Step-1 options together: --keep / --extract (files made here), --phenoColList, --covarColList, --strict,
    --setl0 / --setl1 grids, --cv 3, --use-relative-path -- the driver's files against the oracle's."""
    E = example_dir
    fam = [ln.split() for ln in open(os.path.join(E, "example.fam")).read().split("\n") if ln]
    bim = [ln.split() for ln in open(os.path.join(E, "example.bim")).read().split("\n") if ln]
    keep = str(tmp_path / "keep.txt")
    with open(keep, "w") as fh:
        for t in fam[:420]:
            fh.write("%s %s\n" % (t[0], t[1]))
    ext = str(tmp_path / "extract.txt")
    with open(ext, "w") as fh:
        for j, t in enumerate(bim):
            if j % 5 != 3:
                fh.write(t[1] + "\n")
    args = ["--step", "1", "--bed", os.path.join(E, "example"), "--phenoFile", os.path.join(E, "phenotype_bin_wNA.txt"),
            "--covarFile", os.path.join(E, "covariates.txt"), "--keep", keep, "--extract", ext, "--phenoColList", "Y1,Y2",
            "--covarColList", "V1,V3", "--strict", "--force-qt", "--setl0", "0.1,0.5,0.9", "--setl1", "0.2,0.8", "--cv", "3", "--bsize", "75",
            "--use-relative-path", "--out", "mix"]
    r = _run(args, str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    opt = orc.Step1Options(bed=os.path.join(E, "example"), pheno_file=os.path.join(E, "phenotype_bin_wNA.txt"),
                           covar_file=os.path.join(E, "covariates.txt"), keep=[keep], extract=[ext], pheno_cols=["Y1", "Y2"],
                           covar_cols=["V1", "V3"], strict=True, setl0=[0.1, 0.5, 0.9], setl1=[0.2, 0.8], cv_folds=3, bsize=75,
                           use_rel_path=True, out=str(tmp_path / "omix"))
    ref = orc.run_step1(opt, write_files=True)
    assert open(str(tmp_path / "mix_pred.list")).read() == "Y1 mix_1.loco\nY2 mix_2.loco\n"
    for k in (1, 2):
        h1, ids1, v1, _ = _parse_loco(str(tmp_path / ("mix_%d.loco" % k)))
        h2, ids2, v2, _ = _parse_loco(str(tmp_path / ("omix_%d.loco" % k)))
        assert h1 == h2 and ids1 == ids2 and len(h1) < 420
        assert np.allclose(v1, v2, rtol=2e-5, atol=1e-7, equal_nan=True)
    tab = [ln for ln in r.stdout.split("\n") if " : Rsq = " in ln]
    want = [ln for ln in ref.log if " : Rsq = " in ln]
    assert len(tab) == 4 and [a.split(":")[0].strip() for a in tab] == [b.split(":")[0].strip() for b in want]


def test_cli_pgen_dosages(tmp_path):
    """A .pgen with dosage tracks: regenie switches to PgenReader::Read and analyses non-integer genotypes (Geno.cpp:1101,
    :1795-1822).  The driver decodes them (rg_pgen_read_dosages) and runs level 0 on the fp64 path (rg_l0_blocks_f64); its
    files against the oracle working on the same dosages."""
    from oracle import pgen as opg
    from tests.util import synth_dosages, write_plink
    N, M = 800, 300
    g = synth_dosages(M, N, miss_rate=0.01, seed=21)              # hardcalls 0/1/2, -9 style missing handled by write_plink
    pre = str(tmp_path / "h")
    chroms = np.repeat([1, 2, 3], [100, 100, 100])
    write_plink(pre, g, chroms, P=2, ncov=2, seed=9, missing_pheno=0.02)
    bim = [ln.split() for ln in open(pre + ".bim").read().split("\n") if ln]
    fam = [ln.split() for ln in open(pre + ".fam").read().split("\n") if ln]
    # pgen codes from the bed (so that hardcalls and the companion .bed agree), dosages on most variants in all three layouts
    bedrows = np.fromfile(pre + ".bed", dtype=np.uint8)[3:].reshape(M, (N + 3) // 4)
    codes = np.array([2, 3, 1, 0], dtype=np.uint8)[((bedrows[:, :, None] >> np.array([0, 2, 4, 6])) & 3).reshape(M, -1)[:, :N]]
    rng = np.random.default_rng(3)
    dos = {}
    for j in range(M):
        kind = [None, 0x20, 0x40, 0x60][j % 4]
        if kind is None:
            continue
        k = int(rng.integers(1, (N // 8 if kind == 0x20 else N) + 1))
        ids = np.sort(rng.choice(N, size=k, replace=False))
        hc = np.where(codes[j, ids] == 3, 1.0, codes[j, ids].astype(np.float64))
        vals = np.clip(np.round((hc + rng.normal(0, 0.2, k)) * 16384), 0, 32768).astype(np.uint16)
        dos[j] = (kind, ids, vals)
    pfx = str(tmp_path / "d")
    opg.write_pgen(pfx + ".pgen", codes, [0] * M, wide_vrtypes=True, dosage=dos, reclen_bytes=3)
    with open(pfx + ".pvar", "w") as fh:
        fh.write("#CHROM\tPOS\tID\tREF\tALT\n")
        for t in bim:
            fh.write("%s\t%s\t%s\t%s\t%s\n" % (t[0], t[3], t[1], t[5], t[4]))
    with open(pfx + ".psam", "w") as fh:
        fh.write("#FID\tIID\tSEX\n")
        for t in fam:
            fh.write("%s\t%s\tNA\n" % (t[0], t[1]))
    args = ["--step", "1", "--pgen", pfx, "--phenoFile", pre + ".pheno", "--covarFile", pre + ".covar", "--bsize", "60", "--out", str(tmp_path / "c")]
    r = _run(args, str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "-dosages present: level 0 runs on the fp64 genotype path" in r.stdout
    o = opg.PgenOracle(pfx + ".pgen")
    opt = orc.Step1Options(bed=pre, pheno_file=pre + ".pheno", covar_file=pre + ".covar", bsize=60, out=str(tmp_path / "o"),
                           dosage_provider=lambda offs: np.stack([o.dosages(int(j)) for j in offs]))
    ref = orc.run_step1(opt, write_files=True)
    hard = orc.run_step1(orc.Step1Options(bed=pre, pheno_file=pre + ".pheno", covar_file=pre + ".covar", bsize=60))
    for k in (1, 2):
        h1, ids1, v1, _ = _parse_loco(str(tmp_path / ("c_%d.loco" % k)))
        h2, ids2, v2, _ = _parse_loco(str(tmp_path / ("o_%d.loco" % k)))
        assert h1 == h2 and ids1 == ids2
        assert np.allclose(v1, v2, rtol=2e-5, atol=1e-7, equal_nan=True)
        assert not np.allclose(ref.loco[k - 1], hard.loco[k - 1], rtol=1e-3)      # the dosages matter: hardcalls give other numbers
    # leave-one-out CV on the same dosages
    r = _run(args[:-1] + [str(tmp_path / "cl"), "--loocv"], str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    opt.loocv, opt.out = True, str(tmp_path / "ol")
    orc.run_step1(opt, write_files=True)
    for k in (1, 2):
        _, _, v1, _ = _parse_loco(str(tmp_path / ("cl_%d.loco" % k)))
        _, _, v2, _ = _parse_loco(str(tmp_path / ("ol_%d.loco" % k)))
        assert np.allclose(v1, v2, rtol=2e-5, atol=1e-7, equal_nan=True)


def test_cli_bgen_equals_bed(example_dir, tmp_path):
    """`--bgen example.bgen` (zlib) / example_3chr_zstd.bgen (zstd, + --sample) hold the genotypes of the .bed files: level 0
    runs on the fp64 dosage path there and on the 2-bit path here, and both must give the oracle's LOCO predictions -- two
    independent device routes (the reference's tests likewise compare its bgen and bed runs, test/test_bash.sh:143-216)."""
    E = example_dir
    common = ["--step", "1", "--covarFile", os.path.join(E, "covariates.txt"), "--phenoFile", os.path.join(E, "phenotype.txt"),
              "--remove", os.path.join(E, "fid_iid_to_remove.txt"), "--exclude", os.path.join(E, "snplist_rm.txt"), "--bsize", "100"]
    for bed, bgen, extra in (("example", "example.bgen", []),
                             ("example_3chr", "example_3chr_zstd.bgen", ["--sample", os.path.join(E, "example_3chr.sample")])):
        r = _run(common + ["--bed", os.path.join(E, bed), "--out", str(tmp_path / "b")], str(tmp_path))
        assert r.returncode == 0, r.stdout + r.stderr
        r = _run(common + ["--bgen", os.path.join(E, bgen), "--out", str(tmp_path / "g")] + extra, str(tmp_path))
        assert r.returncode == 0, r.stdout + r.stderr
        assert " * bgen" in r.stdout and "8-bit encoding" in r.stdout
        for k in (1, 2):
            h1, ids1, v1, _ = _parse_loco(str(tmp_path / ("b_%d.loco" % k)))
            h2, ids2, v2, _ = _parse_loco(str(tmp_path / ("g_%d.loco" % k)))
            assert h1 == h2 and ids1 == ids2
            assert np.allclose(v1, v2, rtol=1e-5, atol=1e-7, equal_nan=True)
    # --ref-first flips the counted allele in both readers alike
    r = _run(common + ["--bed", os.path.join(E, "example"), "--ref-first", "--out", str(tmp_path / "br")], str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    r = _run(common + ["--bgen", os.path.join(E, "example.bgen"), "--ref-first", "--out", str(tmp_path / "gr")], str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    _, _, v1, _ = _parse_loco(str(tmp_path / "br_1.loco"))
    _, _, v2, _ = _parse_loco(str(tmp_path / "gr_1.loco"))
    assert np.allclose(v1, v2, rtol=1e-5, atol=1e-7, equal_nan=True)


# ---- one node, several GPUs: the C++ driver's rank threads, the level-0 hand-off and the sharded / shared level 1 ----------
def _world(n, *extra):
    """--gpus n on ONE device over peer copies (what a single-GPU box can run); RG_TEST_REAL_GPUS=1 (tools/first_multigpu.sh, a multi-GPU
    node): n real devices over RCCL."""
    if os.environ.get("RG_TEST_REAL_GPUS") == "1":
        return ["--gpus", str(n)] + list(extra)
    return ["--gpus", str(n), "--single-device", "--transport", "peer"] + list(extra)


def _run_pair(E, tmp_path, extra_common, variants):
    """runs the plain single-GPU command and each variant; returns {name: {file: bytes}}"""
    out = {}
    for name, extra in [("plain", [])] + variants:
        d = tmp_path / name
        d.mkdir()
        r = _run(extra_common + extra + ["--out", "o"], str(d))
        assert r.returncode == 0, name + "\n" + r.stdout[-3000:] + r.stderr[-3000:]
        out[name] = {fn: open(str(d / fn), "rb").read() for fn in sorted(os.listdir(str(d))) if fn.endswith(".loco")}
        out[name]["_log"] = r.stdout
    return out


def test_cli_multi_gpu_qt(example_dir, tmp_path):
    """--gpus 2 on ONE device (peer-copy transport): block ranges as write_l0_master, all-to-all by phenotype + level 1 per
    rank, and the all-gather form with the shared level 1 (--l1-shared) -- .loco files byte-identical to the single-GPU run;
    --gpus 1 --force-collectives runs the same code over RCCL with a world of one (send/recv to self, broadcast, all-reduce)."""
    E = example_dir
    common = ["--step", "1", "--bed", os.path.join(E, "example_3chr"), "--phenoFile", os.path.join(E, "phenotype.txt"),
              "--covarFile", os.path.join(E, "covariates.txt"), "--bsize", "100"]
    res = _run_pair(E, tmp_path, common, [
        ("peer2", _world(2)),
        ("peer2_shared", _world(2, "--l1-shared")),
        ("peer3", _world(3)),
        ("rccl1", ["--gpus", "1", "--force-collectives"]),
        ("rccl1_shared", ["--gpus", "1", "--force-collectives", "--l1-shared"]),
    ])
    for name in ("peer2", "peer2_shared", "peer3", "rccl1", "rccl1_shared"):
        assert sorted(k for k in res[name] if k != "_log") == ["o_1.loco", "o_2.loco"]
        for fn in ("o_1.loco", "o_2.loco"):
            assert res[name][fn] == res["plain"][fn], (name, fn)
    assert "GPU 1 : blocks [4..6]" in res["peer2"]["_log"] and "RCCL" in res["rccl1"]["_log"]


def test_cli_multi_gpu_t2e(tmp_path):
    """--t2e on two and three ranks (one device, peer copies): each trait's Cox ridge runs on the rank that owns it (two ranks), or on
    rank 0 after the all-gather (three ranks for two traits) -- .loco files byte-identical to the single-GPU run, named by the
    reference's column numbers."""
    from tests.test_reference_pin import synth_t2e_case
    _, pre = synth_t2e_case(tmp_path)
    common = ["--step", "1", "--bed", pre, "--phenoFile", pre + ".t2e", "--covarFile", pre + ".covar", "--bsize", "100", "--t2e",
              "--phenoColList", "T1,T2", "--eventColList", "E1,E2"]
    res = _run_pair(None, tmp_path, common, [("peer2", _world(2)), ("peer3", _world(3))])
    for name in ("peer2", "peer3"):
        assert sorted(k for k in res[name] if k != "_log") == ["o_1.loco", "o_3.loco"]
        for fn in ("o_1.loco", "o_3.loco"):
            assert res[name][fn] == res["plain"][fn], (name, fn)
    assert "Deviance = " in res["peer2"]["_log"] and "level 1 of phenotypes [2..2]" in res["peer2"]["_log"]


def test_cli_bt_kfold_quasi_newton_gram_equals_fp64_gram(tmp_path, monkeypatch):
    """The K-fold logistic ridge forms its IRLS Hessians on the 16-bit matrix cores (wgram_bf16.hip: one fp16 operand plane, exact fp64 score and
    stopping rule); RG_WGRAM_F64=1 keeps the fp64 Gram of round 3.  Same fixed point, same stopping rule: the selected ridge value is the
    same, the CV table agrees to its printed digits, the .loco values to a few units of the last printed place (most of them identical as text)."""
    from tests.util import synth_dosages, write_plink
    d = str(tmp_path)
    S = os.path.join(d, "synth")
    chroms = [1] * 120 + [2] * 100 + [3] * 80
    write_plink(S, synth_dosages(300, 5200, miss_rate=0.005, seed=21), chroms, P=3, seed=21, binary=True, missing_pheno=0.02)
    common = ["--step", "1", "--bed", S, "--covarFile", S + ".covar", "--phenoFile", S + ".pheno", "--bsize", "20", "--bt", "--out", "o"]
    outs = {}
    monkeypatch.setenv("RG_WGRAM_QUASI_MIN", "0")            # the quasi-Newton Gram whatever the size (default: from 2e11 flop per Gram on)
    for name in ("quasi", "fp64"):
        dd = tmp_path / name
        dd.mkdir()
        if name == "fp64":
            monkeypatch.setenv("RG_WGRAM_F64", "1")
        r = _run(common, str(dd))
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        outs[name] = {k: _parse_loco(str(dd / ("o_%d.loco" % k))) for k in (1, 2, 3)}
        outs[name]["_log"] = [ln for ln in r.stdout.splitlines() if "-logLik/N" in ln]
    assert len(outs["quasi"]["_log"]) == 15 and outs["quasi"]["_log"] == outs["fp64"]["_log"]        # 3 traits x 5 ridge values, printed digits
    for k in (1, 2, 3):
        a, b = outs["quasi"][k], outs["fp64"][k]
        assert np.array_equal(np.isnan(a[2]), np.isnan(b[2]))
        # six significant digits are printed.  Both runs stop at max |score| < 1e-4 -- the fp64 run wherever its last Newton step lands, the
        # quasi-Newton run (steps on stored Hessians included) below 1e-6 -- so the two iterates differ by what that rule leaves open: a few
        # units of the last printed place on 5,200 samples
        ulp = 10.0 ** (np.floor(np.log10(np.maximum(np.abs(b[2]), 1e-300))) - 5)
        assert np.nanmax(np.abs(a[2] - b[2]) / ulp) <= 6.01
        assert np.nanmean(a[2] == b[2]) >= 0.85


def test_cli_t2e_null_model_newton_fallback(tmp_path, monkeypatch):
    """The null Cox model's fall-back (fit_null_cox, Step1_Models.cpp:415-436: the Newton solver of cox_firth.cpp without the Firth term,
    used when the coordinate descent does not converge) fits the same model: forced with RG_COX_NULL_FORCE_NEWTON=1 it gives the same
    penalties, deviances and LOCO predictors as the coordinate descent up to the two solvers' stopping tolerance (score < 2.5e-4)."""
    from tests.test_reference_pin import synth_t2e_case
    _, pre = synth_t2e_case(tmp_path)
    common = ["--step", "1", "--bed", pre, "--phenoFile", pre + ".t2e", "--covarFile", pre + ".covar", "--bsize", "100", "--t2e",
              "--phenoColList", "T1,T2", "--eventColList", "E1,E2", "--out", "o"]
    outs = {}
    for name, env in (("cd", None), ("newton", "1")):
        d = tmp_path / name
        d.mkdir()
        if env:
            monkeypatch.setenv("RG_COX_NULL_FORCE_NEWTON", env)
        r = _run(common, str(d))
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        outs[name] = {fn: _parse_loco(str(d / fn))[2] for fn in ("o_1.loco", "o_3.loco")}
        outs[name]["_log"] = r.stdout
    for fn in ("o_1.loco", "o_3.loco"):
        a, b = outs["cd"][fn], outs["newton"][fn]
        assert np.array_equal(np.isnan(a), np.isnan(b))
        assert np.nanmax(np.abs(a - b)) <= 1e-2 * np.nanmax(np.abs(a))      # the descent stops on a relative objective change of 2.5e-4
    dev = lambda lg: [float(x.split("Deviance = ")[1].split("<")[0]) for x in lg.splitlines() if "Deviance = " in x]
    assert dev(outs["newton"]["_log"]) == pytest.approx(dev(outs["cd"]["_log"]), rel=2e-3)


@pytest.mark.parametrize("form", ["by_phenotype", "all_gather"])
def test_cli_multi_gpu_rank_failure_does_not_hang(example_dir, tmp_path, form):
    """A rank whose host side fails (here: the .bed is cut short, so the reader thread of the LAST rank runs out of file) breaks the
    group (rg_group_abort): the other rank -- waiting in the level-0 exchange -- returns an error too, the process exits with the
    reference's `ERROR:` line and a failure status, nothing is left waiting in a barrier or a receive, and no .loco file is written."""
    import shutil
    E = example_dir
    d = tmp_path / "cut"
    d.mkdir()
    for ext in (".bim", ".fam"):
        shutil.copy(os.path.join(E, "example_3chr" + ext), str(d / ("g" + ext)))
    raw = open(os.path.join(E, "example_3chr.bed"), "rb").read()
    open(str(d / "g.bed"), "wb").write(raw[:len(raw) - 4000])           # the last variants are missing
    cmd = ["--step", "1", "--bed", str(d / "g"), "--phenoFile", os.path.join(E, "phenotype.txt"), "--covarFile", os.path.join(E, "covariates.txt"),
           "--bsize", "100"] + _world(2) + ["--out", "o"]
    if form == "all_gather":
        cmd.append("--l1-shared")
    r = subprocess.run([BIN] + cmd, capture_output=True, text=True, cwd=str(d), timeout=120)    # a hang fails the test by timeout
    assert r.returncode != 0
    assert "ERROR:" in r.stdout and "cannot read bed file" in r.stdout, r.stdout[-2000:]
    assert not [fn for fn in os.listdir(str(d)) if fn.endswith(".loco")]


def test_cli_multi_gpu_bt_and_loocv(example_dir, tmp_path):
    E = example_dir
    common = ["--step", "1", "--bed", os.path.join(E, "example"), "--phenoFile", os.path.join(E, "phenotype_bin.txt"),
              "--covarFile", os.path.join(E, "covariates.txt"), "--remove", os.path.join(E, "fid_iid_to_remove.txt"),
              "--exclude", os.path.join(E, "snplist_rm.txt"), "--bsize", "100", "--bt"]
    res = _run_pair(E, tmp_path, common, [
        ("peer2", _world(2)),
        ("peer2_gather", _world(2, "--l1-shared")),   # BT: level 1 on rank 0 after the all-gather
        ("rccl1", ["--gpus", "1", "--force-collectives"]),
    ])
    for name in ("peer2", "peer2_gather", "rccl1"):
        for fn in ("o_1.loco", "o_2.loco"):
            assert res[name][fn] == res["plain"][fn], (name, fn)
        assert "0.4504" in [ln for ln in res[name]["_log"].splitlines() if "min value" in ln][1]


# ---- --step 2 --qt: the driver's single-variant score test against regenie's own output ------------------------------------
@pytest.mark.parametrize("route", ["packed", "dense"])
@pytest.mark.parametrize("case", ["qt_bed_3chr", "qt_bed_3chr_opts"])
def test_cli_step2_qt_against_reference_output(example_dir, tmp_path, case, route):
    """`regenie-amd --step 2 --qt --bed ... --pred <LOCO files written by regenie's own step 1>` against the .regenie files
    regenie v4.1.2 wrote for the same command (tests/golden/ref_outputs/step2/<case>_Y*.regenie.gz, made by oracle/_ref/regenie):
    identifying columns, A1FREQ and N as text, BETA / SE / CHISQ / LOG10P to the printed digits.  `_opts` drops samples (--remove:
    the analysed samples are re-packed), counts the other allele (--ref-first) and raises --minMAC; both library routes -- the
    hard-call i8 route the driver takes and the fp64 route (RG_S2_DENSE=1) -- must meet the same bar and agree on every line."""
    import gzip
    E = example_dir
    R = os.path.join(ROOT, "tests", "golden", "ref_outputs")
    with open(str(tmp_path / "pred.list"), "w") as pl:
        for k in (1, 2):
            fn = str(tmp_path / ("ref_%d.loco" % k))
            open(fn, "wb").write(gzip.open(os.path.join(R, "qt_kfold_3chr", "out_%d.loco.gz" % k), "rb").read())
            pl.write("Y%d %s\n" % (k, fn))
    extra = {"qt_bed_3chr": ["--bsize", "200"],
             "qt_bed_3chr_opts": ["--remove", os.path.join(E, "fid_iid_to_remove.txt"), "--ref-first", "--minMAC", "40", "--bsize", "300"]}[case]
    env = {k: v for k, v in os.environ.items() if k != "RG_S2_DENSE"}
    if route == "dense":
        env["RG_S2_DENSE"] = "1"
    r = subprocess.run([BIN, "--step", "2", "--bed", os.path.join(E, "example_3chr"), "--phenoFile", os.path.join(E, "phenotype.txt"),
                        "--covarFile", os.path.join(E, "covariates.txt"), "--qt", "--pred", str(tmp_path / "pred.list"), "--out", "s2"] + extra,
                       cwd=str(tmp_path), capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    for k in (1, 2):
        got = open(str(tmp_path / ("s2_Y%d.regenie" % k))).read().splitlines()
        ref = gzip.open(os.path.join(R, "step2", "%s_Y%d.regenie.gz" % (case, k)), "rt").read().splitlines()
        assert got[0] == ref[0] and len(got) == len(ref) == (501 if case == "qt_bed_3chr" else 494)
        same = 0
        for a, b in zip(got[1:], ref[1:]):
            ta, tb = a.split(" "), b.split(" ")
            assert ta[:8] == tb[:8] and ta[12] == tb[12] == "NA", (a, b)          # CHROM GENPOS ID ALLELE0 ALLELE1 A1FREQ N TEST ... EXTRA
            for x, y in zip(ta[8:12], tb[8:12]):
                assert float(x) == pytest.approx(float(y), rel=2e-5, abs=2e-9), (a, b)
            same += a == b
        assert same >= 0.9 * (len(ref) - 1), same                                  # most lines byte-identical
    ign = [ln for ln in r.stdout.splitlines() if "ignored tests due to low MAC" in ln]
    assert ign and ign[0].split(":")[1].strip() == ("0" if case == "qt_bed_3chr" else "14")


@pytest.mark.parametrize("mode", ["qt", "bt_firth", "bt_spa_bgen", "gz"])
def test_cli_step2_multi_gpu(example_dir, tmp_path, mode):
    """`--step 2 --gpus 3`: the run's blocks in contiguous ranges on three library contexts (one host thread each; --single-device puts them on
    one GPU, RG_TEST_REAL_GPUS=1 on three), no collective -- every result file byte-identical to the one-GPU run, the count of ignored tests
    and (binary traits) the null-Firth estimates the parts wrote merged into one file per trait."""
    import gzip
    E = example_dir
    R = os.path.join(ROOT, "tests", "golden", "ref_outputs")
    bt = mode.startswith("bt")
    s1 = "bt_loocv_refcmd" if bt else "qt_kfold_3chr"
    with open(str(tmp_path / "pred.list"), "w") as pl:
        for k in (1, 2):
            fn = str(tmp_path / ("ref_%d.loco" % k))
            open(fn, "wb").write(gzip.open(os.path.join(R, s1, "out_%d.loco.gz" % k), "rb").read())
            pl.write("Y%d %s\n" % (k, fn))
    if bt:
        geno = ["--bgen", os.path.join(E, "example.bgen")] if mode == "bt_spa_bgen" else ["--bed", os.path.join(E, "example")]
        cmd = ["--step", "2"] + geno + ["--phenoFile", os.path.join(E, "phenotype_bin.txt"), "--covarFile", os.path.join(E, "covariates.txt"),
               "--remove", os.path.join(E, "fid_iid_to_remove.txt"), "--bsize", "100", "--bt", "--pThresh", "0.05", "--pred", str(tmp_path / "pred.list")]
        cmd += ["--spa"] if mode == "bt_spa_bgen" else ["--firth", "--approx", "--write-null-firth"]
    else:
        cmd = ["--step", "2", "--bed", os.path.join(E, "example_3chr"), "--phenoFile", os.path.join(E, "phenotype.txt"), "--covarFile", os.path.join(E, "covariates.txt"),
               "--qt", "--bsize", "70", "--pred", str(tmp_path / "pred.list")] + (["--gz"] if mode == "gz" else [])
    world = ["--gpus", "3"] + ([] if os.environ.get("RG_TEST_REAL_GPUS") == "1" else ["--single-device"])
    one = subprocess.run([BIN] + cmd + ["--out", "one"], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    three = subprocess.run([BIN] + cmd + world + ["--out", "three"], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert one.returncode == 0 and three.returncode == 0, one.stdout[-2000:] + three.stdout[-3000:] + three.stderr[-2000:]
    suffix = ".regenie.gz" if mode == "gz" else ".regenie"
    for k in (1, 2):
        a = open(str(tmp_path / ("one_Y%d%s" % (k, suffix))), "rb").read()
        b = open(str(tmp_path / ("three_Y%d%s" % (k, suffix))), "rb").read()
        if mode == "gz":
            a, b = gzip.decompress(a), gzip.decompress(b)
        assert a == b and a.count(b"\n") > 400
    assert not [fn for fn in os.listdir(str(tmp_path)) if ".part" in fn]
    ign = lambda r: [ln for ln in r.stdout.splitlines() if "ignored tests due to low MAC" in ln][0]      # noqa: E731
    assert ign(one) == ign(three) and "GPU 2 : blocks" in three.stdout
    if mode == "bt_firth":
        for k in (1, 2):
            assert open(str(tmp_path / ("one_%d.firth" % k))).read() == open(str(tmp_path / ("three_%d.firth" % k))).read()
    if mode == "bt_spa_bgen":
        # the .bgen blocks reach the device through the read-ahead (inflate + one byte walk into 2-byte rows, driver_step2.cpp `prepare`); the
        # general route (three double rows per variant, RG_S2_BGEN_ROWS=1) must print the same files
        rows = subprocess.run([BIN] + cmd + ["--out", "rows"], cwd=str(tmp_path), capture_output=True, text=True, timeout=300, env=dict(os.environ, RG_S2_BGEN_ROWS="1"))
        assert rows.returncode == 0, rows.stdout[-2000:] + rows.stderr[-2000:]
        for k in (1, 2):
            assert open(str(tmp_path / ("one_Y%d.regenie" % k)), "rb").read() == open(str(tmp_path / ("rows_Y%d.regenie" % k)), "rb").read()


@pytest.mark.parametrize("fmt", ["bed", "bgen", "bgen_rf", "pgen", "pgenhc", "strict", "bgen_mininfo"])
def test_cli_step2_qt_masked_phenotypes_against_reference_output(tmp_path, fmt):
    """Phenotypes that differ in their missing values (5 %), genotypes with missing calls (1 %), 3,001 samples x 500 variants x 4
    traits (the synthetic data of the qt_kfold_synth_missing case, regenerated here): regenie takes the sparse branch of
    compute_score_qt for most variants (approximate per-trait denominators); the driver's .regenie files against regenie's own
    (tests/golden/ref_outputs/step2/qt_synth_missing*_Y*.regenie.gz), A1FREQ, INFO and N per trait as text.  Formats: the .bed
    (hard-call i8 route, and the fp64 route with RG_S2_DENSE=1), the same calls as .bgen with 40 % of them smeared into genuine
    8-bit probabilities (INFO < 1; also --ref-first), as .pgen with a 16-bit dosage track (MaCH r2 INFO, check_sparse_G's
    zero-count form) and as a hard-call .pgen."""
    import gzip
    import json
    from tests.util import synth_dosages, write_plink, write_synth_bgen, write_synth_pgen
    R = os.path.join(ROOT, "tests", "golden", "ref_outputs")
    meta = json.load(open(os.path.join(R, "qt_kfold_synth_missing", "meta.json")))
    spec = meta["synthetic"]
    S = str(tmp_path / "synth")
    g = synth_dosages(spec["M"], spec["N"], miss_rate=spec.get("miss_rate", 0.0), seed=spec["seed"])
    write_plink(S, g, spec["chroms"], P=spec["P"], seed=spec["seed"], binary=spec["binary"], missing_pheno=spec["missing_pheno"])
    if fmt.startswith("bgen"):
        write_synth_bgen(S, g, spec["chroms"], seed=spec["seed"])
        src = ["--bgen", S + ".bgen", "--sample", S + ".sample"] + (["--ref-first"] if fmt == "bgen_rf" else []) + (["--minINFO", "0.75"] if fmt == "bgen_mininfo" else [])
    elif fmt.startswith("pgen"):
        write_synth_pgen(S + "_p", g, spec["chroms"], seed=spec["seed"], soft=0.4 if fmt == "pgen" else 0.0)
        src = ["--pgen", S + "_p"]
    else:
        src = ["--bed", S] + (["--strict"] if fmt == "strict" else [])      # --strict: samples with any missing phenotype dropped (2,377 left)
    case = "qt_synth_missing" + ("" if fmt == "bed" else "_" + fmt)
    with open(str(tmp_path / "pred.list"), "w") as pl:
        for k, nm in enumerate(meta["pred_list"]):
            fn = str(tmp_path / ("ref_%d.loco" % (k + 1)))
            open(fn, "wb").write(gzip.open(os.path.join(R, "qt_kfold_synth_missing", "out_%d.loco.gz" % (k + 1)), "rb").read())
            pl.write("%s %s\n" % (nm, fn))
    env = {k: v for k, v in os.environ.items() if k != "RG_S2_DENSE"}
    args = [BIN, "--step", "2"] + src + ["--phenoFile", S + ".pheno", "--covarFile", S + ".covar", "--bsize", "200", "--qt",
                                         "--pred", str(tmp_path / "pred.list"), "--out", "s2"]
    r = subprocess.run(args, cwd=str(tmp_path), capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    ncol = 14 if fmt in ("bgen", "bgen_rf", "pgen", "bgen_mininfo") else 13          # with the INFO column
    for k in range(1, spec["P"] + 1):
        got = open(str(tmp_path / ("s2_Y%d.regenie" % k))).read().splitlines()
        ref = gzip.open(os.path.join(R, "step2", "%s_Y%d.regenie.gz" % (case, k)), "rt").read().splitlines()
        assert got[0] == ref[0] and len(got) == len(ref) and len(ref[0].split(" ")) == ncol
        assert len(ref) == 501 or (fmt == "bgen_mininfo" and 200 < len(ref) < 260)              # --minINFO 0.75 drops about half of the tests, per trait
        same = 0
        for a, b in zip(got[1:], ref[1:]):
            ta, tb = a.split(" "), b.split(" ")
            t0 = ncol - 5                                                          # first of BETA SE CHISQ LOG10P
            assert ta[:5] == tb[:5] and ta[t0 - 2:t0] == tb[t0 - 2:t0] and ta[-1] == tb[-1] == "NA", (a, b)   # ids, N TEST, EXTRA
            for x, y in zip(ta[5:t0 - 2], tb[5:t0 - 2]):                           # A1FREQ (INFO): sums of dosages
                assert x == y or float(x) == pytest.approx(float(y), rel=2e-6), (a, b)
            for x, y in zip(ta[t0:t0 + 4], tb[t0:t0 + 4]):
                assert float(x) == pytest.approx(float(y), rel=2e-5, abs=2e-9), (a, b)
            same += a == b
        assert same >= 0.9 * (len(ref) - 1), same
    if fmt == "bgen_mininfo":
        assert "Number of ignored tests due to low MAC or info score : 1102" in r.stdout
    if fmt not in ("bed", "bgen", "pgen"):
        return
    # the fp64 route of the library (RG_S2_DENSE=1: decoded hard calls / the dosages as doubles instead of the i8 digit routes) must print
    # the same files
    keep = {k: open(str(tmp_path / ("s2_Y%d.regenie" % k))).read() for k in range(1, spec["P"] + 1)}
    r2 = subprocess.run(args, cwd=str(tmp_path), capture_output=True, text=True, timeout=300, env=dict(env, RG_S2_DENSE="1"))
    assert r2.returncode == 0, r2.stdout[-3000:] + r2.stderr[-3000:]
    for k in range(1, spec["P"] + 1):
        a, b = keep[k].splitlines(), open(str(tmp_path / ("s2_Y%d.regenie" % k))).read().splitlines()
        assert len(a) == len(b)
        for la, lb in zip(a[1:], b[1:]):
            ta, tb = la.split(" "), lb.split(" ")
            t0 = ncol - 5
            assert ta[:t0] == tb[:t0]
            for x, y in zip(ta[t0:t0 + 4], tb[t0:t0 + 4]):
                assert float(x) == pytest.approx(float(y), rel=1e-5, abs=1e-9), (la, lb)
    if fmt != "bgen":
        return
    # RG_S2_BGEN_HOST_SHARE: the host threads decode the last blocks of every group beside the device decoder -- the same files, byte for byte
    import re
    a3 = list(args)
    a3[a3.index("--bsize") + 1] = "40"
    r3 = subprocess.run(a3, cwd=str(tmp_path), capture_output=True, text=True, timeout=300, env=dict(env, RG_S2_BGEN_HOST_SHARE="0.5", RG_TIMING="1"))
    assert r3.returncode == 0, r3.stdout[-3000:] + r3.stderr[-3000:]
    m = re.search(r"BGEN on the device: (\d+) blocks \((\d+) on the host route\)", r3.stderr)
    assert m and int(m.group(1)) > 0 and int(m.group(2)) > 0, r3.stderr[-2000:]
    for k in range(1, spec["P"] + 1):
        assert open(str(tmp_path / ("s2_Y%d.regenie" % k))).read() == keep[k]


def test_cli_step2_bt_score_test_against_reference_output(example_dir, tmp_path):
    """`regenie-amd --step 2 --bt` (the score test without Firth / SPA: null logistic model with the LOCO offset per chromosome,
    compute_score_bt, get_sumstats) against regenie's own output for the documented Step-2 command without --firth
    (tests/golden/ref_outputs/step2/bt_score_bed_Y*.regenie.gz; --remove drops samples, so the rows are re-packed)."""
    import gzip
    E = example_dir
    R = os.path.join(ROOT, "tests", "golden", "ref_outputs")
    with open(str(tmp_path / "pred.list"), "w") as pl:
        for k in (1, 2):
            fn = str(tmp_path / ("ref_%d.loco" % k))
            open(fn, "wb").write(gzip.open(os.path.join(R, "bt_loocv_refcmd", "out_%d.loco.gz" % k), "rb").read())
            pl.write("Y%d %s\n" % (k, fn))
    r = _run(["--step", "2", "--bed", os.path.join(E, "example"), "--phenoFile", os.path.join(E, "phenotype_bin.txt"),
              "--covarFile", os.path.join(E, "covariates.txt"), "--remove", os.path.join(E, "fid_iid_to_remove.txt"), "--bsize", "200", "--bt",
              "--pred", str(tmp_path / "pred.list"), "--out", "s2"], str(tmp_path))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    for k in (1, 2):
        got = open(str(tmp_path / ("s2_Y%d.regenie" % k))).read().splitlines()
        ref = gzip.open(os.path.join(R, "step2", "bt_score_bed_Y%d.regenie.gz" % k), "rt").read().splitlines()
        assert got[0] == ref[0] and len(got) == len(ref) > 900
        same = 0
        for a, b in zip(got[1:], ref[1:]):
            ta, tb = a.split(" "), b.split(" ")
            assert ta[:8] == tb[:8] and ta[12] == tb[12] == "NA", (a, b)
            for x, y in zip(ta[8:12], tb[8:12]):
                assert float(x) == pytest.approx(float(y), rel=2e-5, abs=2e-9), (a, b)
            same += a == b
        assert same >= 0.9 * (len(ref) - 1), same


def test_cli_step2_bt_approx_firth_reproduces_the_reference_held_golden(example_dir, tmp_path):
    """The documented Step-2 command of the reference (docs/docs/options.md:36-53: --step 2 --bgen example.bgen --bt --firth --approx
    --pThresh 0.01 --remove ...) fed by regenie's own Step-1 LOCO files: Y1 against example/test_bin_out_firth_Y1.regenie, the one
    golden output the reference holds, Y2 against the output of oracle/_ref/regenie.  Binary-trait score test on 8-bit .bgen dosages
    (integer route of the contraction primitive) + null Firth model per chromosome + the 1-parameter Firth fit of the ~3 % of tests
    whose |z| exceeds the threshold.  Rows without the correction to the printed digits; corrected rows to regenie's own stopping
    tolerance (its build and its golden file differ by 2e-5 there)."""
    import gzip
    E = example_dir
    R = os.path.join(ROOT, "tests", "golden", "ref_outputs")
    with open(str(tmp_path / "pred.list"), "w") as pl:
        for k in (1, 2):
            fn = str(tmp_path / ("ref_%d.loco" % k))
            open(fn, "wb").write(gzip.open(os.path.join(R, "bt_loocv_refcmd", "out_%d.loco.gz" % k), "rb").read())
            pl.write("Y%d %s\n" % (k, fn))
    r = _run(["--step", "2", "--bgen", os.path.join(E, "example.bgen"), "--phenoFile", os.path.join(E, "phenotype_bin.txt"),
              "--covarFile", os.path.join(E, "covariates.txt"), "--remove", os.path.join(E, "fid_iid_to_remove.txt"), "--bsize", "200", "--bt",
              "--firth", "--approx", "--pThresh", "0.01", "--pred", str(tmp_path / "pred.list"), "--out", "s2"], str(tmp_path))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    refs = {1: open(os.path.join(E, "test_bin_out_firth_Y1.regenie")).read().splitlines(),
            2: gzip.open(os.path.join(R, "step2", "bt_firth_bgen_Y2.regenie.gz"), "rt").read().splitlines()}
    plain = {k: gzip.open(os.path.join(R, "step2", "bt_score_bed_Y%d.regenie.gz" % k), "rt").read().splitlines() for k in (1, 2)}
    ncorr = 0
    for k in (1, 2):
        got = open(str(tmp_path / ("s2_Y%d.regenie" % k))).read().splitlines()
        ref = refs[k]
        assert got[0] == ref[0] and len(got) == len(ref) == 1001
        same = 0
        for a, b in zip(got[1:], ref[1:]):
            ta, tb = a.split(" "), b.split(" ")
            assert ta[:9] == tb[:9] and ta[13] == tb[13] == "NA", (a, b)          # CHROM GENPOS ID ALLELE0 ALLELE1 A1FREQ INFO N TEST ... EXTRA
            corrected = float(tb[11]) > 6.6                                        # CHISQ above the 0.99 quantile: a Firth row
            ncorr += corrected
            for x, y in zip(ta[9:13], tb[9:13]):
                assert float(x) == pytest.approx(float(y), rel=2e-4 if corrected else 2e-5, abs=2e-9), (a, b)
            same += a == b
        assert same >= 900, same
    assert 20 <= ncorr <= 40


def test_cli_step2_bt_exact_firth_against_reference_output(example_dir, tmp_path):
    """`--firth` without `--approx` (the covariates refitted with every flagged variant: fit_firth_logistic_snp, null fit under the full
    penalty, then the full fit) on the documented command: against the output of oracle/_ref/regenie
    (tests/golden/ref_outputs/step2/bt_firth_exact_bgen_Y*.regenie.gz), 29 corrected tests."""
    import gzip
    E = example_dir
    R = os.path.join(ROOT, "tests", "golden", "ref_outputs")
    with open(str(tmp_path / "pred.list"), "w") as pl:
        for k in (1, 2):
            fn = str(tmp_path / ("ref_%d.loco" % k))
            open(fn, "wb").write(gzip.open(os.path.join(R, "bt_loocv_refcmd", "out_%d.loco.gz" % k), "rb").read())
            pl.write("Y%d %s\n" % (k, fn))
    r = _run(["--step", "2", "--bgen", os.path.join(E, "example.bgen"), "--phenoFile", os.path.join(E, "phenotype_bin.txt"),
              "--covarFile", os.path.join(E, "covariates.txt"), "--remove", os.path.join(E, "fid_iid_to_remove.txt"), "--bsize", "200", "--bt",
              "--firth", "--pThresh", "0.01", "--pred", str(tmp_path / "pred.list"), "--out", "s2"], str(tmp_path))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    loose = 0
    for k in (1, 2):
        got = open(str(tmp_path / ("s2_Y%d.regenie" % k))).read().splitlines()
        ref = gzip.open(os.path.join(R, "step2", "bt_firth_exact_bgen_Y%d.regenie.gz" % k), "rt").read().splitlines()
        assert got[0] == ref[0] and len(got) == len(ref) == 1001
        for a, b in zip(got[1:], ref[1:]):
            ta, tb = a.split(" "), b.split(" ")
            assert ta[:9] == tb[:9] and ta[13] == tb[13] == "NA", (a, b)
            if not all(float(x) == pytest.approx(float(y), rel=2e-5, abs=2e-9) for x, y in zip(ta[9:13], tb[9:13])):
                loose += 1                                                          # a corrected row: regenie's stopping tolerance
                for x, y in zip(ta[9:13], tb[9:13]):
                    assert float(x) == pytest.approx(float(y), rel=3e-4, abs=2e-9), (a, b)
    assert loose <= 29


def test_cli_step2_bt_spa_against_reference_output(example_dir, tmp_path):
    """`--spa` on the documented command (.bgen dosages): saddlepoint p-values of the 29 tests above the threshold, run_SPA_test_snp with its
    Newton iteration restated step for step -- every row to the printed digits (tests/golden/ref_outputs/step2/bt_spa_bgen_Y*.regenie.gz)."""
    import gzip
    E = example_dir
    R = os.path.join(ROOT, "tests", "golden", "ref_outputs")
    with open(str(tmp_path / "pred.list"), "w") as pl:
        for k in (1, 2):
            fn = str(tmp_path / ("ref_%d.loco" % k))
            open(fn, "wb").write(gzip.open(os.path.join(R, "bt_loocv_refcmd", "out_%d.loco.gz" % k), "rb").read())
            pl.write("Y%d %s\n" % (k, fn))
    r = _run(["--step", "2", "--bgen", os.path.join(E, "example.bgen"), "--phenoFile", os.path.join(E, "phenotype_bin.txt"),
              "--covarFile", os.path.join(E, "covariates.txt"), "--remove", os.path.join(E, "fid_iid_to_remove.txt"), "--bsize", "200", "--bt",
              "--spa", "--pThresh", "0.01", "--pred", str(tmp_path / "pred.list"), "--out", "s2"], str(tmp_path))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    for k in (1, 2):
        got = open(str(tmp_path / ("s2_Y%d.regenie" % k))).read().splitlines()
        ref = gzip.open(os.path.join(R, "step2", "bt_spa_bgen_Y%d.regenie.gz" % k), "rt").read().splitlines()
        assert got[0] == ref[0] and len(got) == len(ref) == 1001
        same = 0
        for a, b in zip(got[1:], ref[1:]):
            ta, tb = a.split(" "), b.split(" ")
            assert ta[:9] == tb[:9] and ta[13] == tb[13], (a, b)
            for x, y in zip(ta[9:13], tb[9:13]):
                assert float(x) == pytest.approx(float(y), rel=3e-5, abs=2e-9), (a, b)
            same += a == b
        assert same >= 900, same


def test_cli_step2_bt_spa_rare_variants_against_reference_output(tmp_path):
    """`--spa --pThresh 0.3` on the rare, sparse variants of the Firth test below: 271 corrected tests, the sparse ones in regenie's fast form
    (exact terms for the carriers, normal approximation for the rest), one of them a TEST_FAIL in regenie's output and here."""
    import gzip
    import json
    import shutil
    from tests.util import synth_dosages, synth_rare_dosages, write_bed_bim, write_plink
    R = os.path.join(ROOT, "tests", "golden", "ref_outputs")
    meta = json.load(open(os.path.join(R, "bt_kfold_synth", "meta.json")))
    spec = meta["synthetic"]
    S = str(tmp_path / "synth")
    write_plink(S, synth_dosages(spec["M"], spec["N"], miss_rate=spec.get("miss_rate", 0.0), seed=spec["seed"]), spec["chroms"], P=spec["P"],
                seed=spec["seed"], binary=spec["binary"], missing_pheno=spec["missing_pheno"])
    write_bed_bim(S + "_rare", synth_rare_dosages(300, spec["N"], seed=spec["seed"], miss_rate=0.002), [1] * 100 + [2] * 100 + [5] * 100)
    shutil.copy(S + ".fam", S + "_rare.fam")
    with open(str(tmp_path / "pred.list"), "w") as pl:
        for k, nm in enumerate(meta["pred_list"]):
            fn = str(tmp_path / ("ref_%d.loco" % (k + 1)))
            open(fn, "wb").write(gzip.open(os.path.join(R, "bt_kfold_synth", "out_%d.loco.gz" % (k + 1)), "rb").read())
            pl.write("%s %s\n" % (nm, fn))
    r = _run(["--step", "2", "--bed", S + "_rare", "--phenoFile", S + ".pheno", "--covarFile", S + ".covar", "--bsize", "100", "--bt", "--spa",
              "--pThresh", "0.3", "--pred", str(tmp_path / "pred.list"), "--out", "s2"], str(tmp_path))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    nfail = 0
    for k in range(1, spec["P"] + 1):
        got = open(str(tmp_path / ("s2_Y%d.regenie" % k))).read().splitlines()
        ref = gzip.open(os.path.join(R, "step2", "bt_spa_rare_Y%d.regenie.gz" % k), "rt").read().splitlines()
        assert got[0] == ref[0] and len(got) == len(ref) == 301
        for a, b in zip(got[1:], ref[1:]):
            ta, tb = a.split(" "), b.split(" ")
            assert ta[:8] == tb[:8] and ta[12] == tb[12], (a, b)                   # EXTRA: NA or TEST_FAIL
            nfail += tb[12] == "TEST_FAIL"
            for x, y in zip(ta[8:12], tb[8:12]):
                assert (x == y == "NA") or float(x) == pytest.approx(float(y), rel=3e-5, abs=2e-9), (a, b)
    assert nfail == 1


def test_cli_step2_bt_approx_firth_rare_variants_against_reference_output(tmp_path):
    """Rare, sparse variants (MAF 0.1 - 1 %, 5,200 samples, 3 binary traits with 2 % missing values, --pThresh 0.3): ~270 of the 900 tests get
    the Firth correction, about half of them in regenie's carriers-only form (MAC < 50).  Driver output against regenie's own
    (tests/golden/ref_outputs/step2/bt_firth_rare_Y*.regenie.gz).  Corrected rows: regenie stops at |modified score| < 2.5e-4, a few times
    2.5e-4 * SE^2 from the root, and takes its LRT one iteration before its BETA."""
    import gzip
    import json
    import shutil
    from tests.util import synth_dosages, synth_rare_dosages, write_bed_bim, write_plink
    R = os.path.join(ROOT, "tests", "golden", "ref_outputs")
    meta = json.load(open(os.path.join(R, "bt_kfold_synth", "meta.json")))
    spec = meta["synthetic"]
    S = str(tmp_path / "synth")
    write_plink(S, synth_dosages(spec["M"], spec["N"], miss_rate=spec.get("miss_rate", 0.0), seed=spec["seed"]), spec["chroms"], P=spec["P"],
                seed=spec["seed"], binary=spec["binary"], missing_pheno=spec["missing_pheno"])
    write_bed_bim(S + "_rare", synth_rare_dosages(300, spec["N"], seed=spec["seed"], miss_rate=0.002), [1] * 100 + [2] * 100 + [5] * 100)
    shutil.copy(S + ".fam", S + "_rare.fam")
    with open(str(tmp_path / "pred.list"), "w") as pl:
        for k, nm in enumerate(meta["pred_list"]):
            fn = str(tmp_path / ("ref_%d.loco" % (k + 1)))
            open(fn, "wb").write(gzip.open(os.path.join(R, "bt_kfold_synth", "out_%d.loco.gz" % (k + 1)), "rb").read())
            pl.write("%s %s\n" % (nm, fn))
    r = _run(["--step", "2", "--bed", S + "_rare", "--phenoFile", S + ".pheno", "--covarFile", S + ".covar", "--bsize", "100", "--bt", "--firth", "--approx",
              "--pThresh", "0.3", "--pred", str(tmp_path / "pred.list"), "--out", "s2"], str(tmp_path))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    ncorr = 0
    for k in range(1, spec["P"] + 1):
        got = open(str(tmp_path / ("s2_Y%d.regenie" % k))).read().splitlines()
        ref = gzip.open(os.path.join(R, "step2", "bt_firth_rare_Y%d.regenie.gz" % k), "rt").read().splitlines()
        assert got[0] == ref[0] and len(got) == len(ref) == 301
        for a, b in zip(got[1:], ref[1:]):
            ta, tb = a.split(" "), b.split(" ")
            assert ta[:8] == tb[:8] and ta[12] == tb[12] == "NA", (a, b)
            beta, se, chisq = float(tb[8]), float(tb[9]), float(tb[10])
            strict = all(float(x) == pytest.approx(float(y), rel=2e-5, abs=2e-9) for x, y in zip(ta[8:12], tb[8:12]))
            if not strict:          # a corrected row (they cannot be told from the file: the LRT may fall below the threshold the score test exceeded)
                ncorr += 1
                assert abs(float(ta[8]) - beta) <= 8e-4 * se * se + 2e-5 * abs(beta) + 5e-6, (a, b)
                assert float(ta[9]) == pytest.approx(se, rel=2e-4) and float(ta[10]) == pytest.approx(chisq, rel=2e-3, abs=2e-5), (a, b)
    assert ncorr <= 280          # regenie corrected 271 tests; every other row agrees to the printed digits


def test_cli_step2_ct_score_test_against_reference_output(tmp_path):
    """`regenie-amd --step 2 --ct` (null Poisson model with the LOCO offset per chromosome, compute_score_ct) against regenie's own output
    on synthetic counts: 1,500 samples x 300 variants x 2 traits, 3 % missing phenotypes, 1 % missing calls
    (tests/golden/ref_outputs/step2/ct_synth_Y*.regenie.gz; the LOCO files are regenie's --qt predictions on the same file)."""
    import gzip
    import json
    from tests.util import synth_dosages, write_plink
    R = os.path.join(ROOT, "tests", "golden", "ref_outputs")
    meta = json.load(open(os.path.join(R, "ct_synth", "meta.json")))
    spec = meta["synthetic"]
    S = str(tmp_path / "synth")
    write_plink(S, synth_dosages(spec["M"], spec["N"], miss_rate=spec.get("miss_rate", 0.0), seed=spec["seed"]), spec["chroms"], P=spec["P"],
                seed=spec["seed"], binary=spec["binary"], missing_pheno=spec["missing_pheno"], counts=True)
    with open(str(tmp_path / "pred.list"), "w") as pl:
        for k, nm in enumerate(meta["pred_list"]):
            fn = str(tmp_path / ("ref_%d.loco" % (k + 1)))
            open(fn, "wb").write(gzip.open(os.path.join(R, "ct_synth", "out_%d.loco.gz" % (k + 1)), "rb").read())
            pl.write("%s %s\n" % (nm, fn))
    r = _run(["--step", "2", "--bed", S, "--phenoFile", S + ".pheno", "--covarFile", S + ".covar", "--bsize", "100", "--ct",
              "--pred", str(tmp_path / "pred.list"), "--out", "s2"], str(tmp_path))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    for k in (1, 2):
        got = open(str(tmp_path / ("s2_Y%d.regenie" % k))).read().splitlines()
        ref = gzip.open(os.path.join(R, "step2", "ct_synth_Y%d.regenie.gz" % k), "rt").read().splitlines()
        assert got[0] == ref[0] and len(got) == len(ref) == 301
        same = 0
        for a, b in zip(got[1:], ref[1:]):
            ta, tb = a.split(" "), b.split(" ")
            assert ta[:8] == tb[:8] and ta[12] == tb[12] == "NA", (a, b)
            for x, y in zip(ta[8:12], tb[8:12]):
                assert float(x) == pytest.approx(float(y), rel=2e-5, abs=2e-9), (a, b)
            same += a == b
        assert same >= 270, same


def test_cli_step2_refuses_what_is_not_built(example_dir, tmp_path):
    E = example_dir
    base = ["--step", "2", "--bed", os.path.join(E, "example"), "--phenoFile", os.path.join(E, "phenotype_bin.txt"), "--bsize", "200", "--pred", "x", "--out", "s2"]
    r = _run(base + ["--bt", "--spa", "--firth", "--approx"], str(tmp_path))
    assert r.returncode != 0 and "cannot use both" in r.stdout
    # chromosome X with male samples: the sex-aware allele counts of the non-PAR region are not built -- an error, not different numbers
    import gzip
    import shutil
    for ext in (".bed", ".bim", ".fam"):
        shutil.copy(os.path.join(E, "example_3chr" + ext), str(tmp_path / ("x" + ext)))
    bim = open(str(tmp_path / "x.bim")).read().splitlines()
    open(str(tmp_path / "x.bim"), "w").write("\n".join(("23" + ln[ln.index("\t"):] if ln.split("\t")[0] == "3" else ln) for ln in bim) + "\n")
    fam = open(str(tmp_path / "x.fam")).read().splitlines()
    t = fam[0].split()
    t[4] = "1"
    open(str(tmp_path / "x.fam"), "w").write("\n".join([" ".join(t)] + fam[1:]) + "\n")
    R = os.path.join(ROOT, "tests", "golden", "ref_outputs")
    with open(str(tmp_path / "pred.list"), "w") as pl:
        for k in (1, 2):
            fn = str(tmp_path / ("ref_%d.loco" % k))
            open(fn, "wb").write(gzip.open(os.path.join(R, "qt_kfold_3chr", "out_%d.loco.gz" % k), "rb").read())
            pl.write("Y%d %s\n" % (k, fn))
    r = _run(["--step", "2", "--bed", str(tmp_path / "x"), "--phenoFile", os.path.join(E, "phenotype.txt"), "--covarFile", os.path.join(E, "covariates.txt"),
              "--bsize", "200", "--qt", "--pred", str(tmp_path / "pred.list"), "--out", "s2x"], str(tmp_path))
    assert r.returncode != 0 and "non-PAR" in r.stdout, r.stdout[-2000:]
