"""The C++ driver on the GPU against regenie ITSELF: `regenie-amd --step 1 <args>` must write what regenie v4.1.2 wrote for
the same command (tests/golden/ref_outputs/, produced by oracle/_ref/regenie = the reference's sources compiled by
oracle/Makefile; generator tests/golden/make_ref_outputs.py).  Compared: every .loco / .prs value at the resolution of
the reference's 6-digit text (and under BASELINE.json's max-relative-error metric, bar 1e-5), the NA pattern, sample and
chromosome order, the CV table of the log with the selected ridge parameter, the phenotype names of _pred.list.
Two cases at the full sample count of BASELINE configs[2] / [3] (500,000 samples, 10 QT / 6 BT phenotypes, bsize 1000) run the
driver against the numpy oracle (pinned to regenie by tests/test_reference_pin.py); with RG_LIVE_REFERENCE_500K=1 the
reference binary itself runs next to the driver (minutes per case; last run recorded in profiles/r2_reference_500k.md)."""
import gzip
import json
import os
import subprocess
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
pytest.importorskip("torch")

from tests.test_reference_pin import (EX, REF_OUT, REGENIE, T2E_RE, assert_text_equal, parse_table, read_loco_gz)  # noqa: E402
from tests.golden.make_ref_outputs import CASES, table_lines  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "regenie_amd", "bin", "regenie-amd")


def _loco_file(path):
    lines = open(path).read().splitlines()
    ids = lines[0].split()[1:]
    return ids, np.array([[np.nan if t == "NA" else float(t) for t in ln.split()[1:]] for ln in lines[1:]]), \
        [ln.split()[0] for ln in lines[1:]]


def _loco_file_fast(path):
    """as _loco_file for files with millions of values: one C-level parse per row"""
    with open(path) as fh:
        ids = fh.readline().split()[1:]
        rows, first = [], []
        for ln in fh:
            sp = ln.index(" ")
            first.append(ln[:sp])
            rows.append(np.fromstring(ln[sp + 1:].replace("NA", "nan"), sep=" "))
    return ids, np.array(rows), first


def _compare_tables(got_lines, ref_lines, what, rel=2e-5):
    gt, rt = parse_table(got_lines), parse_table(ref_lines)
    assert len(gt) == len(rt), what
    for ph, (g, r) in enumerate(zip(gt, rt)):
        assert len(g) == len(r), what
        for (h, rsq, mse, ll, mn), (h2, rsq2, mse2, ll2, mn2) in zip(r, g):
            assert h == h2 and mn == mn2, (what, ph, h)
            assert rsq2 == pytest.approx(rsq, rel=rel) and mse2 == pytest.approx(mse, rel=rel, nan_ok=True), (what, ph, h)
            if ll is not None:
                assert ll2 == pytest.approx(ll, rel=rel), (what, ph, h)


@pytest.mark.parametrize("name", sorted(CASES))
def test_driver_reproduces_reference_outputs(name, tmp_path):
    args, spec = CASES[name]
    d = str(tmp_path)
    S = os.path.join(d, "synth")
    if spec:
        from tests.util import synth_dosages, write_plink, write_t2e_pheno
        g = synth_dosages(spec["M"], spec["N"], miss_rate=spec.get("miss_rate", 0.0), seed=spec["seed"])
        write_plink(S, g, spec["chroms"], P=spec["P"], seed=spec["seed"], binary=spec["binary"], missing_pheno=spec["missing_pheno"],
                    counts=spec.get("counts", False))
        if spec.get("t2e"):
            write_t2e_pheno(S + ".t2e", g, seed=spec["seed"], **spec["t2e"])
    r = subprocess.run([BIN] + [a.format(E=EX, S=S) for a in args] + ["--out", "out"], cwd=d, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    meta = json.load(open(os.path.join(REF_OUT, name, "meta.json")))
    assert [ln.split()[0] for ln in open(os.path.join(d, "out_pred.list"))] == meta["pred_list"]
    if "--t2e" in args:      # --t2e: penalties and held-out deviances (Data.cpp:1043-1049), files numbered by the time column
        got_t, ref_t = table_lines(open(os.path.join(d, "out.log")).read()), meta["table"]
        assert len(got_t) == len(ref_t)
        for a, b in zip(got_t, ref_t):
            if b.startswith("phenotype"):
                assert a.split() == b.split()
                continue
            ma, mb = T2E_RE.match(a), T2E_RE.match(b)
            assert ma and mb, (a, b)
            assert float(ma.group(1)) == pytest.approx(float(mb.group(1)), rel=2e-5) and float(ma.group(2)) == pytest.approx(float(mb.group(2)), rel=2e-5)
            assert bool(ma.group(3)) == bool(mb.group(3)), (a, b)
    else:
        _compare_tables(table_lines(open(os.path.join(d, "out.log")).read()), meta["table"], name)
    n = 0
    for fn in sorted(os.listdir(os.path.join(REF_OUT, name))):
        if not (fn.endswith(".loco.gz") or fn.endswith(".prs.gz")):
            continue
        ids_r, ref = read_loco_gz(os.path.join(REF_OUT, name, fn))
        ids_g, got, first = _loco_file(os.path.join(d, fn[:-3]))
        assert ids_g == ids_r, fn
        if fn.endswith(".loco.gz"):
            assert first == [str(c) for c in range(1, 24)]
        assert_text_equal(got, ref, "%s %s" % (name, fn))
        n += 1
    assert n >= len(meta["pred_list"])


def _read_firth(path):
    op = gzip.open if path.endswith(".gz") else open
    return [(ln.split()[0], [float(t) for t in ln.split()[1:]]) for ln in op(path, "rt").read().splitlines()]


def _firth_equal_up_to_basis_signs(got, ref):
    """The estimates are coefficients in the orthonormal covariate basis (getBasis: eigenvectors of X^T X), whose columns are defined up to
    sign: the eigen-solvers of the two programs may pick different ones, the same way on every line of every file of a run."""
    sign = np.sign(np.array(got[0][1]) * np.array(ref[0][1]))
    assert np.all(sign != 0)
    for (_, a), (_, b) in zip(got, ref):
        assert np.array(a) * sign == pytest.approx(np.array(b), rel=3e-3, abs=3e-4)


def test_driver_write_and_use_null_firth(tmp_path):
    """--write-null-firth in Step 1 (Data.cpp:1873-1902): out_<k>.firth = per chromosome the covariate estimates of the null approximate-Firth
    model with that chromosome's LOCO prediction as offset, out_firth.list names them -- against the files regenie wrote for the same command
    (the bt_kfold_synth case; 1e-3 relative was seen until round 5 and put down to regenie's stopping rule -- it was the masked samples, which regenie keeps in
    X^T W X with weight 1 (driver_models.cpp firth_fit_cols; tests/test_firth_null_cpu.py): the oracle with that behaviour is within 7e-5 of these files, 9e-4 without; 3e-3 is still
    what this test allows).  Then
    `--step 2 --firth --approx --use-null-firth LIST --write-null-firth` on the rare variants: the stored estimates are start values only, so every
    result line equals the run without them to the stopping tolerance of the null fit, and the estimates Step 2 writes are those of its own null Firth fits."""
    from tests.util import synth_dosages, synth_rare_dosages, write_bed_bim, write_plink
    import shutil
    args, spec = CASES["bt_kfold_synth"]
    assert "--write-null-firth" in args
    d = str(tmp_path)
    S = os.path.join(d, "synth")
    write_plink(S, synth_dosages(spec["M"], spec["N"], miss_rate=spec.get("miss_rate", 0.0), seed=spec["seed"]), spec["chroms"], P=spec["P"], seed=spec["seed"],
                binary=spec["binary"], missing_pheno=spec["missing_pheno"])
    r = subprocess.run([BIN] + [a.format(E=EX, S=S) for a in args] + ["--out", "out"], cwd=d, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "List of files with null Firth estimates written to" in r.stdout
    lst = [ln.split() for ln in open(os.path.join(d, "out_firth.list"))]
    assert [t[0] for t in lst] == ["Y1", "Y2", "Y3"] and all(os.path.isabs(t[1]) and t[1].endswith("out_%d.firth" % (k + 1)) for k, t in enumerate(lst))
    for k in (1, 2, 3):
        ref = _read_firth(os.path.join(REF_OUT, "bt_kfold_synth", "out_%d.firth.gz" % k))
        got = _read_firth(os.path.join(d, "out_%d.firth" % k))
        assert [c for c, _ in got] == [c for c, _ in ref] == [str(c) for c in range(1, 24)]
        _firth_equal_up_to_basis_signs(got, ref)
    # step 2 with and without the stored estimates
    write_bed_bim(S + "_rare", synth_rare_dosages(300, spec["N"], seed=spec["seed"], miss_rate=0.002), [1] * 100 + [2] * 100 + [5] * 100)
    shutil.copy(S + ".fam", S + "_rare.fam")
    base = [BIN, "--step", "2", "--bed", S + "_rare", "--covarFile", S + ".covar", "--phenoFile", S + ".pheno", "--bsize", "100", "--bt", "--firth", "--approx",
            "--pThresh", "0.3", "--pred", os.path.join(d, "out_pred.list")]
    r0 = subprocess.run(base + ["--out", "plain"], cwd=d, capture_output=True, text=True, timeout=600)
    r1 = subprocess.run(base + ["--use-null-firth", os.path.join(d, "out_firth.list"), "--write-null-firth", "--out", "warm"], cwd=d, capture_output=True, text=True, timeout=600)
    assert r0.returncode == 0 and r1.returncode == 0, r0.stdout[-2000:] + r1.stdout[-2000:]
    assert "reading null Firth estimates using file" in r1.stdout
    for k in (1, 2, 3):
        a = open(os.path.join(d, "plain_Y%d.regenie" % k)).read().splitlines()
        b = open(os.path.join(d, "warm_Y%d.regenie" % k)).read().splitlines()
        ref = gzip.open(os.path.join(REF_OUT, "step2", "bt_firth_rare_usenull_Y%d.regenie.gz" % k), "rt").read().splitlines()
        assert len(a) == len(b) == len(ref) and a[0] == b[0] == ref[0]
        for x, y, z in zip(a[1:], b[1:], ref[1:]):
            tx, ty, tz = x.split(), y.split(), z.split()
            assert tx[:8] == ty[:8] == tz[:8] and tx[12] == ty[12] == tz[12]
            if "NA" in tx[8:12] + ty[8:12] + tz[8:12]:
                assert tx[8:12] == ty[8:12] == tz[8:12]
                continue
            # the start values do not move the maximisers -- but since round 5 the null Firth fit stops where regenie's stops (|modified score| < 50 numtol from the
            # second iteration on), so two starts end a tolerance apart and the corrected rows with them: the bars are those of the comparison with regenie below
            # (twice those bars: each run is within one of them of regenie's)
            b0, se0 = float(tx[8]), float(tx[9])
            assert abs(float(ty[8]) - b0) <= 1.6e-3 * se0 * se0 + 4e-5 * abs(b0) + 1e-5, (x, y)
            assert float(ty[9]) == pytest.approx(se0, rel=4e-4) and float(ty[10]) == pytest.approx(float(tx[10]), rel=4e-3, abs=4e-5), (x, y)
            assert float(ty[11]) == pytest.approx(float(tx[11]), rel=4e-3, abs=4e-5), (x, y)
            # regenie's own run with --use-null-firth.  Its corrected rows stop at |modified score| < 2.5e-4 (a few times 2.5e-4 * SE^2 from the
            # root) and take the LRT one iteration before BETA: the bounds of test_cli_step2_bt_approx_firth_rare_variants_against_reference_output
            beta, se, chisq, logp = (float(t) for t in tz[8:12])
            assert abs(float(ty[8]) - beta) <= 8e-4 * se * se + 2e-5 * abs(beta) + 5e-6, (y, z)
            assert float(ty[9]) == pytest.approx(se, rel=2e-4) and float(ty[10]) == pytest.approx(chisq, rel=2e-3, abs=2e-5), (y, z)
            assert float(ty[11]) == pytest.approx(logp, rel=2e-3, abs=2e-5), (y, z)
        # what Step 2 wrote: the chromosomes it tested, the estimates of its own null Firth fits (= regenie's, to its stopping tolerance)
        got = _read_firth(os.path.join(d, "warm_%d.firth" % k))
        ref2 = _read_firth(os.path.join(REF_OUT, "step2", "bt_firth_rare_usenull_%d.firth.gz" % k))
        assert [c for c, _ in got] == [c for c, _ in ref2] == ["1", "2", "5"]
        _firth_equal_up_to_basis_signs(got, ref2)


def _write_big(prefix, N, M, chroms, P, binary, seed, missing_pheno):
    """N x M .bed with HWE genotypes (MAF U(0.05, 0.5), 0.2 % missing calls), P phenotypes with a polygenic signal."""
    rng = np.random.default_rng(seed)
    maf = 0.05 + 0.45 * rng.random(M)
    score = np.zeros((P, N))
    with open(prefix + ".bed", "wb") as fh:
        fh.write(b"\x6c\x1b\x01")
        for j in range(M):
            g = (rng.random(N) < maf[j]).astype(np.int8) + (rng.random(N) < maf[j]).astype(np.int8)
            gs = (g - 2 * maf[j]) / np.sqrt(2 * maf[j] * (1 - maf[j]))
            for p in range(P):
                if (j + p) % 7 == 0:
                    score[p] += gs * (0.05 if (j // 7) % 2 else -0.04)
            code = np.where(g == 2, 0, np.where(g == 1, 2, 3)).astype(np.uint8)
            code[rng.random(N) < 0.002] = 1
            c = code.reshape(-1, 4)
            fh.write((c[:, 0] | (c[:, 1] << 2) | (c[:, 2] << 4) | (c[:, 3] << 6)).astype(np.uint8).tobytes())
    with open(prefix + ".bim", "w") as fh:
        for j in range(M):
            fh.write("%d\ts%d\t0\t%d\tA\tG\n" % (chroms[j], j, j + 1))
    with open(prefix + ".fam", "w") as fh:
        fh.write("".join("%d %d 0 0 0 -9\n" % (i + 1, i + 1) for i in range(N)))
    cov = rng.standard_normal((N, 2))
    with open(prefix + ".covar", "w") as fh:
        fh.write("FID IID C1 C2\n")
        fh.write("".join("%d %d %.10g %.10g\n" % (i + 1, i + 1, cov[i, 0], cov[i, 1]) for i in range(N)))
    Y = score.T / score.std(axis=1) * 0.5 + rng.standard_normal((N, P)) + 0.3 * cov[:, :1]
    if binary:
        Y = (Y > np.quantile(Y, 0.8, axis=0, keepdims=True)).astype(np.float64)
    miss = rng.random((N, P)) < missing_pheno
    with open(prefix + ".pheno", "w") as fh:
        fh.write("FID IID " + " ".join("Y%d" % (p + 1) for p in range(P)) + "\n")
        for i in range(N):
            fh.write("%d %d " % (i + 1, i + 1) + " ".join("NA" if miss[i, p] else "%.10g" % Y[i, p] for p in range(P)) + "\n")


def _big_case(kind, d):
    N = 500_000
    if kind == "qt_config3_shape":
        M, chroms, P, binary = 1400, [1] * 1000 + [2] * 400, 10, False
    else:
        M, chroms, P, binary = 1300, [3] * 1000 + [9] * 300, 6, True
    pre = os.path.join(d, "big")
    _write_big(pre, N, M, chroms, P, binary, seed=2026, missing_pheno=0.01)
    common = ["--step", "1", "--bed", pre, "--covarFile", pre + ".covar", "--phenoFile", pre + ".pheno", "--bsize", "1000",
              "--bt" if binary else "--qt"]
    return N, P, binary, pre, common


def _check_vals(got, ref, what):
    assert ref.shape[0] == 23 and got.shape == ref.shape
    assert np.array_equal(np.isnan(ref), np.isnan(got)), what
    ok = ~np.isnan(ref)
    return float(np.max(np.abs(got[ok] - ref[ok])) / np.max(np.abs(ref[ok]))), ok


def _compare_loco_files(d, P, kind, ref_prefix):
    worst = 0.0
    for ph in range(1, P + 1):
        ids_r, ref, _ = _loco_file_fast(os.path.join(d, "%s_%d.loco" % (ref_prefix, ph)))
        ids_g, got, _ = _loco_file_fast(os.path.join(d, "gpu_%d.loco" % ph))
        assert ids_r == ids_g
        e, ok = _check_vals(got, ref, (kind, ph))
        # both files carry 6 significant digits: one unit of the last printed digit + the metric of BASELINE.json
        mag = np.maximum(np.maximum(np.abs(ref[ok]), np.abs(got[ok])), 1e-300)
        ulp = 10.0 ** (np.floor(np.log10(mag)) - 5)
        assert float(np.max(np.abs(got[ok] - ref[ok]) / ulp)) <= 1.0 + 1e-6, (kind, ph)
        worst = max(worst, e)
    assert worst < 1e-5, worst
    return worst


@pytest.mark.parametrize("kind", ["qt_config3_shape", "bt_kfold_config4_shape"])
def test_driver_vs_oracle_at_500k_samples(kind, tmp_path):
    """BASELINE configs[2] / [3] at their real sample count (500,000), phenotype count (10 QT with missing values / 6 BT,
    K-fold) and block size 1000, on two SNP blocks (one full, one chromosome end): the C++ driver against the numpy oracle
    -- which tests/test_reference_pin.py pins to regenie itself -- reading the same files.  LOCO predictors at the
    resolution of the driver's 6-digit text (bar 1e-5, BASELINE.json), CV tables, selected ridge parameters."""
    from oracle import regenie_step1 as orc
    from tests.test_reference_pin import oracle_loco_rows
    d = str(tmp_path)
    N, P, binary, pre, common = _big_case(kind, d)
    g = subprocess.run([BIN] + common + ["--out", "gpu"], cwd=d, capture_output=True, text=True, timeout=1200)
    assert g.returncode == 0, g.stdout[-3000:] + g.stderr[-3000:]
    t0 = time.time()
    res = orc.run_step1(orc.Step1Options(bed=pre, pheno_file=pre + ".pheno", covar_file=pre + ".covar", bsize=1000, bt=binary), keep_W=False)
    assert not res.use_loocv
    print("%s: oracle %.1f s" % (kind, time.time() - t0))
    _compare_tables(table_lines(open(os.path.join(d, "gpu.log")).read()),
                    [l for l in res.log if l.startswith("phenotype ") or ": Rsq = " in l], kind)
    worst = 0.0
    for ph in range(P):
        ids_g, got, first = _loco_file_fast(os.path.join(d, "gpu_%d.loco" % (ph + 1)))
        ids_o, ref = oracle_loco_rows(res, ph)
        assert ids_g == ids_o and first == [str(c) for c in range(1, 24)]
        e, ok = _check_vals(got, ref, (kind, ph))
        mag = np.maximum(np.abs(got[ok]), 1e-300)       # the text's own decade (0.0999996 prints as 0.1)
        ulp = 10.0 ** (np.floor(np.log10(mag)) - 5)
        # the text rounds the driver's value (half a unit) and the driver agrees with the oracle to ~1e-8 (QT) / ~1e-6 (the
        # iterative logistic fits) OF THE COLUMN'S SCALE: within one unit of the last printed digit, except for entries that are
        # tiny against the scale (a LOCO value of 1e-8 carries the absolute rounding of sums of order 1); then BASELINE's metric
        tol = np.maximum(ulp, 1e-9 * np.max(np.abs(ref[ok])))
        assert float(np.max(np.abs(got[ok] - ref[ok]) / tol)) <= 1.0, (kind, ph)
        worst = max(worst, e)
    assert worst < 1e-5, worst
    print("%s: LOCO max-rel-err vs the oracle (6-digit text) %.2e" % (kind, worst))


def test_driver_t2e_vs_oracle_at_500k_samples(tmp_path):
    """`--t2e` at 500,000 samples (two traits, two SNP blocks, bsize 1000): event times rounded to 0.001 -- about twenty tied events per
    distinct time -- and 1 % of the (time, event) pairs missing.  The device's risk-set scans then run over chunks of ~490 samples per
    thread (tests/test_l1_cox_gpu.py has two), the held-out folds hold 100,000 samples each.  Driver against the oracle
    (oracle/regenie_step1_t2e.py, pinned to regenie by tests/test_reference_pin.py): penalties, deviances, the selected penalty and
    the LOCO predictors at the resolution of the 6-digit text."""
    from oracle import regenie_step1 as orc
    from oracle import regenie_step1_t2e as t2e
    from tests.test_reference_pin import T2E_RE
    d = str(tmp_path)
    N = 500_000
    pre = os.path.join(d, "big")
    _write_big(pre, N, 1100, [4] * 1000 + [11] * 100, 2, False, seed=2027, missing_pheno=0.0)
    ph = np.loadtxt(pre + ".pheno", skiprows=1, usecols=(2, 3))
    rng = np.random.default_rng(5)
    with open(pre + ".t2e", "w") as fh:
        fh.write("FID IID T1 E1 T2 E2\n")
        cols = []
        for q in range(2):
            lp = 0.4 * (ph[:, q] - ph[:, q].mean()) / ph[:, q].std()
            t_ev, t_c = rng.exponential(1.0, N) * np.exp(-lp) * (4.0 + q), rng.exponential(7.0, N)
            tm, ev = np.round(np.minimum(t_ev, t_c), 3) + 0.001, (t_ev <= t_c).astype(int)
            cols.append((tm, ev, rng.random(N) < 0.01))
        fh.write("".join("%d %d %s\n" % (i + 1, i + 1, " ".join("NA NA" if m[i] else "%.3f %d" % (t[i], e[i]) for t, e, m in cols)) for i in range(N)))
    common = ["--step", "1", "--bed", pre, "--covarFile", pre + ".covar", "--phenoFile", pre + ".t2e", "--bsize", "1000", "--t2e",
              "--phenoColList", "T1,T2", "--eventColList", "E1,E2"]
    g = subprocess.run([BIN] + common + ["--out", "gpu"], cwd=d, capture_output=True, text=True, timeout=1200)
    assert g.returncode == 0, g.stdout[-3000:] + g.stderr[-3000:]
    t0 = time.time()
    res = t2e.run_step1_t2e(orc.Step1Options(bed=pre, pheno_file=pre + ".t2e", covar_file=pre + ".covar", bsize=1000), {"T1": "E1", "T2": "E2"})
    print("t2e at 500k: oracle %.1f s" % (time.time() - t0))
    got_t = table_lines(open(os.path.join(d, "gpu.log")).read())
    ref_t = [ln.rstrip() for ln in res["log"]]
    assert len(got_t) == len(ref_t) == 12
    for a, b in zip(got_t, ref_t):
        if b.startswith("phenotype"):
            assert a.split() == b.split()
            continue
        ma, mb = T2E_RE.match(a), T2E_RE.match(b)
        assert float(ma.group(1)) == pytest.approx(float(mb.group(1)), rel=2e-5) and float(ma.group(2)) == pytest.approx(float(mb.group(2)), rel=2e-5), (a, b)
        assert bool(ma.group(3)) == bool(mb.group(3)), (a, b)
    prep = res["prep"]
    order = [i for i in sorted(range(len(prep.ids)), key=lambda i: prep.ids[i]) if prep.ind_in_analysis[i]]
    for tn, num in (("T1", 1), ("T2", 3)):
        ti = prep.pheno_names.index(tn)
        ids_g, got, first = _loco_file_fast(os.path.join(d, "gpu_%d.loco" % num))
        ref = res["traits"][tn]["loco"][order, :].T.copy()
        ref[:, ~prep.mask[order, ti]] = np.nan
        assert ids_g == [prep.ids[i] for i in order] and first == [str(c) for c in range(1, 24)]
        e, ok = _check_vals(got, ref, ("t2e", tn))
        mag = np.maximum(np.abs(got[ok]), 1e-300)
        tol = np.maximum(10.0 ** (np.floor(np.log10(mag)) - 5), 1e-7 * np.max(np.abs(ref[ok])))
        assert float(np.max(np.abs(got[ok] - ref[ok]) / tol)) <= 1.0, tn
        assert e < 1e-5


needs_ref_binary = pytest.mark.skipif(not os.path.exists(REGENIE), reason="oracle/_ref/regenie not built (make -C oracle)")


@needs_ref_binary
@pytest.mark.skipif(os.environ.get("RG_LIVE_REFERENCE_500K") != "1",
                    reason="regenie itself at 500,000 samples takes 5-10 minutes per case on the GPU box's host (its Eigen GEMM on 255 "
                           "threads); set RG_LIVE_REFERENCE_500K=1.  Last run: profiles/r2_reference_500k.md")
@pytest.mark.parametrize("kind", ["qt_config3_shape", "bt_kfold_config4_shape"])
def test_driver_vs_live_reference_at_500k_samples(kind, tmp_path):
    """The same two cases with regenie ITSELF (oracle/_ref/regenie) run next to the driver on the same files."""
    d = str(tmp_path)
    N, P, binary, pre, common = _big_case(kind, d)
    t0 = time.time()
    r = subprocess.run([REGENIE] + common + ["--out", "ref"], cwd=d, capture_output=True, text=True, timeout=3000)
    t_ref = time.time() - t0
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    t0 = time.time()
    g = subprocess.run([BIN] + common + ["--out", "gpu"], cwd=d, capture_output=True, text=True, timeout=1200)
    t_gpu = time.time() - t0
    assert g.returncode == 0, g.stdout[-3000:] + g.stderr[-3000:]
    print("%s: reference %.1f s, driver %.1f s (wall, from files)" % (kind, t_ref, t_gpu))
    _compare_tables(table_lines(open(os.path.join(d, "gpu.log")).read()), table_lines(open(os.path.join(d, "ref.log")).read()), kind)
    print("%s: LOCO max-rel-err vs regenie %.2e" % (kind, _compare_loco_files(d, P, kind, "ref")))


@pytest.mark.skipif(not os.path.exists(REGENIE), reason="oracle/_ref/regenie not built")
def test_driver_against_live_reference_on_drawn_cases(tmp_path, monkeypatch):
    """Cases drawn by tests/golden/fuzz_oracle_vs_reference.py (QT / BT, K-fold / leave-one-out, sizes, block size, folds, grid sizes, --ref-first,
    --strict, missing genotypes / phenotypes): regenie itself runs beside `regenie-amd --step 1` and `--step 2 --qt` on the same inputs; the
    driver's files are held to regenie's (1e-5 of the largest value; the round's 80-case run, byte-identical throughout: tests/golden/fuzz_driver_log.md)."""
    from tests.golden import fuzz_oracle_vs_reference as fz
    monkeypatch.setenv("FUZZ_DRIVER", "1")
    for seed in (1, 2, 3, 4, 6):
        line, ok = fz.run_one(seed, str(tmp_path))
        assert ok and "driver: loco" in line, line
