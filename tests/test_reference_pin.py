"""Pins the oracle to regenie ITSELF (CPU, no GPU).

tests/golden/ref_outputs/ holds outputs of regenie v4.1.2 compiled from /root/reference by oracle/Makefile
(oracle/_ref/regenie; generating script tests/golden/make_ref_outputs.py).  Here the numpy restatement
(oracle/regenie_step1.py, oracle/regenie_step2_qt.py) must reproduce them:

 * every Step-1 route -- QT K-fold (BASELINE configs[0]), QT LOOCV, BT LOOCV (the reference's own test command),
   BT K-fold (needs >= 5,000 samples: synthetic), missing genotypes / phenotypes, --remove / --cv 3 / --ref-first /
   --print-prs -- on the .loco text (6 significant digits => 1e-5 of the largest value is the bar of BASELINE.json,
   the comparison here is per value at the text's own resolution), the CV tables of the log and the selected ridge value;
 * the level-0 predictors at full fp64 precision through the reference's --split-l0/--run-l0/--keep-l0 job files;
 * the build itself: when oracle/_ref/regenie is present it must reproduce the reference-HELD golden file
   example/test_bin_out_firth_Y1.regenie (docs/docs/options.md:20-53) and the committed fixtures byte for byte.
"""
import gzip
import json
import os
import shutil
import re
import subprocess

import numpy as np
import pytest

from oracle import regenie_step1 as orc

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_OUT = os.path.join(HERE, "golden", "ref_outputs")
EX = os.path.join(HERE, "golden", "example")
REGENIE = os.path.join(ROOT, "oracle", "_ref", "regenie")


def read_loco_gz(path):
    """-> (ids, values [nchrom][n] with NaN for NA)"""
    with gzip.open(path, "rt") as fh:
        lines = fh.read().splitlines()
    ids = lines[0].split()[1:]
    rows = []
    for ln in lines[1:]:
        tok = ln.split()
        rows.append([np.nan if t == "NA" else float(t) for t in tok[1:]])
    return ids, np.array(rows)


def oracle_loco_rows(res, ph):
    """The oracle's LOCO matrix in the file's layout: std::map id order over analysed samples, NaN where masked."""
    prep = res.prep
    order = [i for i in sorted(range(len(prep.ids)), key=lambda i: prep.ids[i]) if prep.ind_in_analysis[i]]
    v = res.loco[ph][order, :].T.copy()
    v[:, ~prep.mask[order, ph]] = np.nan
    return [prep.ids[i] for i in order], v


def assert_text_equal(got, ref, what):
    """`ref` was printed with 6 significant digits: got must round to it (half an ulp of the text + fp64 slack)."""
    assert got.shape == ref.shape, what
    assert np.array_equal(np.isnan(got), np.isnan(ref)), what + ": NA pattern"
    ok = ~np.isnan(ref)
    mag = np.maximum(np.abs(ref[ok]), 1e-300)
    ulp = 10.0 ** (np.floor(np.log10(mag)) - 5)
    err = np.abs(got[ok] - ref[ok])
    worst = float(np.max(err / ulp)) if err.size else 0.0
    assert worst <= 0.5 + 1e-3, "%s: %.3f text ulps" % (what, worst)
    # BASELINE.json's metric as well
    assert float(np.max(err)) / float(np.max(np.abs(ref[ok]))) < 1e-5, what


TABLE_RE = re.compile(r"^\s*(\S+)\s*: Rsq = ([^,]+)(?:, MSE = ([^,<]+))?(?:, -logLik/N = ([^<]+))?(<- min value)?")     # --ct prints no MSE


def parse_table(lines):
    """-> list per phenotype of [(h, rsq, mse, ll or None, is_min)]"""
    out = []
    for ln in lines:
        if ln.startswith("phenotype "):
            out.append([])
            continue
        m = TABLE_RE.match(ln)
        assert m, ln
        out[-1].append((float(m.group(1)), float(m.group(2)), float(m.group(3)) if m.group(3) else float("nan"),
                        float(m.group(4)) if m.group(4) else None, bool(m.group(5))))
    return out


def check_case(name, opt, synth=None, tmp_path=None):
    meta = json.load(open(os.path.join(REF_OUT, name, "meta.json")))
    if synth is not None:
        from tests.util import synth_dosages, write_plink
        spec = meta["synthetic"]
        pre = str(tmp_path / "synth")
        g = synth_dosages(spec["M"], spec["N"], miss_rate=spec.get("miss_rate", 0.0), seed=spec["seed"])
        write_plink(pre, g, spec["chroms"], P=spec["P"], seed=spec["seed"], binary=spec["binary"],
                    missing_pheno=spec["missing_pheno"], counts=spec.get("counts", False))
        opt.bed, opt.pheno_file, opt.covar_file = pre, pre + ".pheno", pre + ".covar"
    res = orc.run_step1(opt)
    ref_tab = parse_table(meta["table"])
    got_tab = parse_table([l for l in res.log if l.startswith("phenotype ") or ": Rsq = " in l])
    assert len(ref_tab) == len(got_tab) == len(meta["pred_list"])
    for ph, (rt, gt) in enumerate(zip(ref_tab, got_tab)):
        assert len(rt) == len(gt)
        for (h, rsq, mse, ll, mn), (h2, rsq2, mse2, ll2, mn2) in zip(rt, gt):
            assert h == h2 and mn == mn2, (name, ph, h)
            assert rsq2 == pytest.approx(rsq, rel=2e-5) and mse2 == pytest.approx(mse, rel=2e-5, nan_ok=True), (name, ph, h)
            if ll is not None:
                assert ll2 == pytest.approx(ll, rel=2e-5), (name, ph, h)
        ids, ref = read_loco_gz(os.path.join(REF_OUT, name, "out_%d.loco.gz" % (ph + 1)))
        gids, got = oracle_loco_rows(res, ph)
        assert ids == gids, name
        assert_text_equal(got, ref, "%s pheno %d" % (name, ph + 1))
    return res


def E(*p):
    return os.path.join(EX, *p)


def test_qt_kfold_config1():
    check_case("qt_kfold_config1", orc.Step1Options(bed=E("example"), pheno_file=E("phenotype.txt"),
                                                    covar_file=E("covariates.txt"), bsize=100))


def test_qt_kfold_3chr():
    check_case("qt_kfold_3chr", orc.Step1Options(bed=E("example_3chr"), pheno_file=E("phenotype.txt"),
                                                 covar_file=E("covariates.txt"), bsize=100))


def test_qt_kfold_nb():
    """--nb 4: the first four blocks in chromosome order (one of chromosome 1, three of chromosome 2's four, none of chromosome 3); the
    variants past them are not analysed (set_blocks, Data.cpp:314-329)"""
    check_case("qt_kfold_3chr_nb", orc.Step1Options(bed=E("example_3chr"), pheno_file=E("phenotype.txt"), covar_file=E("covariates.txt"), bsize=100, n_block=4))


def test_qt_kfold_options():
    check_case("qt_kfold_3chr_opts", orc.Step1Options(bed=E("example_3chr"), pheno_file=E("phenotype.txt"),
                                                      covar_file=E("covariates.txt"), remove=[E("fid_iid_to_remove.txt")],
                                                      bsize=70, cv_folds=3, ref_first=True))


def test_qt_loocv():
    check_case("qt_loocv_3chr", orc.Step1Options(bed=E("example_3chr"), pheno_file=E("phenotype.txt"),
                                                 covar_file=E("covariates.txt"), bsize=100, loocv=True))


def test_bt_loocv_reference_command():
    res = check_case("bt_loocv_refcmd", orc.Step1Options(bed=E("example"), pheno_file=E("phenotype_bin.txt"),
                                                         covar_file=E("covariates.txt"), remove=[E("fid_iid_to_remove.txt")],
                                                         exclude=[E("snplist_rm.txt")], bsize=100, bt=True))
    assert res.use_loocv


def test_bt_loocv_missing_phenotypes():
    check_case("bt_loocv_wNA", orc.Step1Options(bed=E("example_3chr"), pheno_file=E("phenotype_bin_wNA.txt"),
                                                covar_file=E("covariates.txt"), bsize=100, bt=True))


def test_bt_kfold_synthetic(tmp_path):
    res = check_case("bt_kfold_synth", orc.Step1Options(bsize=100, bt=True), synth=True, tmp_path=tmp_path)
    assert not res.use_loocv


def test_ct_kfold_synthetic(tmp_path):
    """--ct Step 1 (ridge_poisson_level_1, Step1_Models.cpp:1429-1580; make_predictions_count, Data.cpp:1575-1622; the penalty grid and
    labels of check_l0 / Data::output for counts) against regenie's own run: Poisson counts with a polygenic rate, on which its
    K-fold fit converges (round 4; before, the count route was pinned by the oracle only)."""
    check_case("ct_kfold_synth", orc.Step1Options(bsize=100, ct=True), synth=True, tmp_path=tmp_path)


def test_qt_kfold_synthetic_missing(tmp_path):
    check_case("qt_kfold_synth_missing", orc.Step1Options(bsize=128), synth=True, tmp_path=tmp_path)


def synth_t2e_case(tmp_path, name="t2e_kfold_synth"):
    """The synthetic time-to-event data set of a fixture: (meta, prefix); phenotype file = prefix + '.t2e'."""
    from tests.util import synth_dosages, write_plink, write_t2e_pheno
    meta = json.load(open(os.path.join(REF_OUT, name, "meta.json")))
    spec = meta["synthetic"]
    pre = str(tmp_path / "synth")
    g = synth_dosages(spec["M"], spec["N"], miss_rate=spec.get("miss_rate", 0.0), seed=spec["seed"])
    write_plink(pre, g, spec["chroms"], P=spec["P"], seed=spec["seed"], binary=spec["binary"], missing_pheno=spec["missing_pheno"])
    write_t2e_pheno(pre + ".t2e", g, seed=spec["seed"], **spec["t2e"])
    return meta, pre


T2E_RE = re.compile(r"^\s*(\S+)\s*: Deviance = ([^<]+)(<- min value)?")


@pytest.mark.parametrize("case,extra", [("t2e_kfold_synth", {}), ("t2e_kfold_synth_event_l0", {"t2e_event_l0": True}),
                                        ("t2e_kfold_synth_pi6", {"t2e_l1_pi6": True})])
def test_t2e_cox_ridge_synthetic(tmp_path, case, extra):
    """`--step 1 --t2e` (oracle/regenie_step1_t2e.py: null Cox model, Cox ridge paths per fold by cyclic coordinate descent, held-out
    deviances, out-of-fold predictions) against regenie's own run: two traits with tied event times and missing pairs; the penalties
    and deviances of the log to its six digits, the same penalty selected, the LOCO files at the text's resolution.  Also with the
    reference's two undocumented level-1 switches, `--t2e-event-l0` (level-0 predictors of the event column) and `--t2e-l1-pi6`
    (penalties from the heritability grid)."""
    from oracle import regenie_step1_t2e as t2e
    meta, pre = synth_t2e_case(tmp_path, case)
    opt = orc.Step1Options(bed=pre, pheno_file=pre + ".t2e", covar_file=pre + ".covar", bsize=100, **extra)
    res = t2e.run_step1_t2e(opt, {"T1": "E1", "T2": "E2"})
    ref_lines = [ln for ln in meta["table"]]
    got_lines = [ln.rstrip() for ln in res["log"]]
    assert [ln for ln in ref_lines if ln.startswith("phenotype")] == [ln.rstrip() for ln in got_lines if ln.startswith("phenotype")] == \
        ["phenotype 1 (T1) :", "phenotype 3 (T2) :"]
    for a, b in zip(ref_lines, got_lines):
        if a.startswith("phenotype"):
            continue
        ma, mb = T2E_RE.match(a), T2E_RE.match(b)
        assert ma and mb, (a, b)
        assert float(mb.group(1)) == pytest.approx(float(ma.group(1)), rel=2e-5) and float(mb.group(2)) == pytest.approx(float(ma.group(2)), rel=2e-5)
        assert bool(ma.group(3)) == bool(mb.group(3)), (a, b)
    prep = res["prep"]
    order = [i for i in sorted(range(len(prep.ids)), key=lambda i: prep.ids[i]) if prep.ind_in_analysis[i]]
    for tn in ("T1", "T2"):
        ti = prep.pheno_names.index(tn)
        ids, ref = read_loco_gz(os.path.join(REF_OUT, case, "out_%d.loco.gz" % (ti + 1)))
        got = res["traits"][tn]["loco"][order, :].T.copy()
        got[:, ~prep.mask[order, ti]] = np.nan
        assert ids == [prep.ids[i] for i in order]
        assert_text_equal(got, ref, "%s %s" % (case, tn))


def test_t2e_cox_ridge_example_with_options():
    """--t2e on the example genotypes with --remove / --cv 3 / --ref-first: time columns out of file order (regenie numbers its outputs 1 and 3),
    pairs missing for one trait or both (the covariates are centred over ALL kept samples, analysed or not, Pheno.cpp:1663-1667)."""
    from oracle import regenie_step1_t2e as t2e
    meta = json.load(open(os.path.join(REF_OUT, "t2e_kfold_3chr_opts", "meta.json")))
    opt = orc.Step1Options(bed=E("example_3chr"), pheno_file=E("phenotype_t2e.txt"), covar_file=E("covariates.txt"), remove=[E("fid_iid_to_remove.txt")],
                           bsize=70, cv_folds=3, ref_first=True)
    res = t2e.run_step1_t2e(opt, {"Relapse_T": "Relapse", "Surv": "Died"})
    got_lines = [ln.rstrip() for ln in res["log"]]
    assert len(got_lines) == len(meta["table"]) and meta["pred_list"] == ["Surv", "Relapse_T"]
    for a, b in zip(meta["table"], got_lines):
        if a.startswith("phenotype"):
            assert a.split() == b.split()
            continue
        ma, mb = T2E_RE.match(a), T2E_RE.match(b)
        assert float(mb.group(1)) == pytest.approx(float(ma.group(1)), rel=2e-5) and float(mb.group(2)) == pytest.approx(float(ma.group(2)), rel=2e-5)
        assert bool(ma.group(3)) == bool(mb.group(3)), (a, b)
    prep = res["prep"]
    order = [i for i in sorted(range(len(prep.ids)), key=lambda i: prep.ids[i]) if prep.ind_in_analysis[i]]
    for tn in ("Surv", "Relapse_T"):
        ti = prep.pheno_names.index(tn)
        ids, ref = read_loco_gz(os.path.join(REF_OUT, "t2e_kfold_3chr_opts", "out_%d.loco.gz" % (ti + 1)))
        got = res["traits"][tn]["loco"][order, :].T.copy()
        got[:, ~prep.mask[order, ti]] = np.nan
        assert ids == [prep.ids[i] for i in order]
        assert_text_equal(got, ref, "t2e_kfold_3chr_opts %s" % tn)


def test_level0_predictors_full_precision():
    """The reference's --run-l0 job files are its level-0 predictors as raw doubles, column-major N x (blocks*R0) per
    phenotype (Step1_Models.cpp:728-734): ridge_level_0 of the oracle against them at fp64 resolution."""
    res = orc.run_step1(orc.Step1Options(bed=E("example_3chr"), pheno_file=E("phenotype.txt"),
                                         covar_file=E("covariates.txt"), bsize=100))
    N = res.prep.Y.shape[0]
    for ph in (0, 1):
        cols = []
        for job in (1, 2):
            raw = np.frombuffer(gzip.open(os.path.join(REF_OUT, "qt_split_l0_3chr", "par_job%d_l0_Y%d.gz" % (job, ph + 1)), "rb").read(),
                                dtype="<f8")
            assert raw.size % N == 0
            cols.append(raw.reshape(-1, N).T)
        Wref = np.concatenate(cols, axis=1)
        assert Wref.shape == res.W[ph].shape
        err = np.max(np.abs(res.W[ph] - Wref)) / np.max(np.abs(Wref))
        assert err < 1e-11, err
    # and the reference's own identity: the split run's .loco equal the single run's, byte for byte (test_bash.sh:126-138)
    for ph in (1, 2):
        a = gzip.open(os.path.join(REF_OUT, "qt_split_l0_3chr", "out_%d.loco.gz" % ph), "rb").read()
        b = gzip.open(os.path.join(REF_OUT, "qt_kfold_3chr", "out_%d.loco.gz" % ph), "rb").read()
        assert a == b


def _read_regenie(path_or_bytes):
    txt = gzip.open(path_or_bytes, "rt").read() if isinstance(path_or_bytes, str) and path_or_bytes.endswith(".gz") \
        else open(path_or_bytes).read()
    lines = txt.splitlines()
    hdr = lines[0].split()
    return hdr, [ln.split() for ln in lines[1:]]


def test_step2_qt_oracle_against_reference():
    """compute_res / residualize_geno / compute_score_qt / get_logp of the oracle (oracle/regenie_step2_qt.py) against the
    reference's Step-2 output on the .bed, fed by the reference's own Step-1 LOCO files (printed with 6 digits)."""
    from oracle import regenie_step2_qt as s2
    opt = orc.Step1Options(bed=E("example_3chr"), pheno_file=E("phenotype.txt"), covar_file=E("covariates.txt"), bsize=100)
    bim, chrom, offs, snp_ids, prep = orc.load_inputs(opt)
    bed, _ = orc.open_bed(opt.bed + ".bed", prep.n_file)
    ia = prep.ind_in_analysis
    ids = [i for i, k in zip(prep.ids, ia) if k]
    n, P = int(ia.sum()), prep.Y.shape[1]
    loco = []
    for ph in range(P):
        hdr, v = read_loco_gz(os.path.join(REF_OUT, "qt_kfold_3chr", "out_%d.loco.gz" % (ph + 1)))
        pos = {s: k for k, s in enumerate(hdr)}
        loco.append(v[:, [pos[i] for i in ids]])                  # [nchrom][n] in the analysis order
    refs = [_read_regenie(os.path.join(REF_OUT, "step2", "qt_bed_3chr_Y%d.regenie.gz" % (ph + 1))) for ph in range(P)]
    col = {nm: i for i, nm in enumerate(refs[0][0])}
    X, Y, mask = prep.X[ia], prep.Y[ia], prep.mask[ia].astype(np.float64)
    row = 0
    for c in sorted(set(chrom.tolist())):
        blup = np.stack([loco[ph][c - 1] for ph in range(P)], axis=1)
        res, _, scf = s2.compute_res(Y, blup * mask, mask, prep.Neff, X.shape[1], prep.scale_Y)
        sel = np.flatnonzero(chrom == c)
        G = orc.decode_bed_rows(np.asarray(bed[offs[sel]]), prep.n_file)[:, ~prep.ind_ignore][:, ia]
        out = s2.score_qt_block(G, X, res, mask, scf)
        for k in range(sel.size):
            for ph in range(P):
                r = refs[ph][1][row + k]
                assert r[col["ID"]] == snp_ids[sel[k]]
                beta, se, chisq, logp = (float(r[col[nm]]) for nm in ("BETA", "SE", "CHISQ", "LOG10P"))
                # the LOCO input was rounded to 6 digits by the writer: the statistics agree to ~1e-5
                assert out["bhat"][k, ph] == pytest.approx(beta, rel=5e-5, abs=2e-6)
                assert out["se"][k, ph] == pytest.approx(se, rel=5e-5)
                assert out["chisq"][k, ph] == pytest.approx(chisq, rel=1e-4, abs=2e-6)
                assert s2.get_logp(out["chisq"][k, ph]) == pytest.approx(logp, rel=1e-4, abs=2e-6)
        row += sel.size
    assert row == len(refs[0][1])


def test_step2_qt_oracle_sparse_branch_against_reference(tmp_path):
    """Phenotypes that differ in their missing values (5 %), genotypes with missing calls (1 %): the reference takes the sparse
    branch of compute_score_qt (approximate per-trait denominators, Step2_Models.cpp:402-413) for the variants check_sparse_G
    flags and the dense branch for the others; oracle s2.score_qt_block_ref against regenie's own output for 500 variants x 4
    traits on the synthetic data of the qt_kfold_synth_missing case (regenerated here, deterministic)."""
    import json
    from oracle import regenie_step2_qt as s2
    from tests.util import synth_dosages, write_plink
    meta = json.load(open(os.path.join(REF_OUT, "qt_kfold_synth_missing", "meta.json")))
    spec = meta["synthetic"]
    S = str(tmp_path / "synth")
    write_plink(S, synth_dosages(spec["M"], spec["N"], miss_rate=spec.get("miss_rate", 0.0), seed=spec["seed"]), spec["chroms"], P=spec["P"],
                seed=spec["seed"], binary=spec["binary"], missing_pheno=spec["missing_pheno"])
    opt = orc.Step1Options(bed=S, pheno_file=S + ".pheno", covar_file=S + ".covar", bsize=128, test_mode=True)
    bim, chrom, offs, snp_ids, prep = orc.load_inputs(opt)
    bed, _ = orc.open_bed(opt.bed + ".bed", prep.n_file)
    ia = prep.ind_in_analysis
    ids = [i for i, k in zip(prep.ids, ia) if k]
    P = prep.Y.shape[1]
    loco = []
    for ph in range(P):
        hdr, v = read_loco_gz(os.path.join(REF_OUT, "qt_kfold_synth_missing", "out_%d.loco.gz" % (ph + 1)))
        pos = {s: k for k, s in enumerate(hdr)}
        loco.append(v[:, [pos[i] for i in ids]])
    refs = [_read_regenie(os.path.join(REF_OUT, "step2", "qt_synth_missing_Y%d.regenie.gz" % (ph + 1))) for ph in range(P)]
    col = {nm: i for i, nm in enumerate(refs[0][0])}
    X, Y, mask = prep.X[ia], prep.Y[ia], prep.mask[ia].astype(np.float64)
    assert (mask.min(axis=0) == 0).all()                       # every trait has missing values
    row, nsparse, ndense, moved = 0, 0, 0, 0
    for c in sorted(set(chrom.tolist())):
        blup = np.stack([loco[ph][c - 1] for ph in range(P)], axis=1)             # the .loco has a line for every chromosome 1..23
        res, _, scf = s2.compute_res(Y, blup * mask, mask, prep.Neff, X.shape[1], prep.scale_Y)
        sel = np.flatnonzero(chrom == c)
        G = orc.decode_bed_rows(np.asarray(bed[offs[sel]]), prep.n_file)[:, ~prep.ind_ignore][:, ia]
        out = s2.score_qt_block_ref(G, X, res, mask, scf, n_samples=int((~prep.ind_ignore).sum()))
        dense = s2.score_qt_block(G, X, res, mask, scf)
        nsparse += int(out["sparse"].sum()); ndense += int((1 - out["sparse"]).sum())
        moved = max(moved, float(np.nanmax(np.abs(out["chisq"] / dense["chisq"] - 1.0))))
        for k in range(sel.size):
            for ph in range(P):
                r = refs[ph][1][row + k]
                assert r[col["ID"]] == snp_ids[sel[k]]
                beta, se, chisq, logp = (float(r[col[nm]]) for nm in ("BETA", "SE", "CHISQ", "LOG10P"))
                assert out["bhat"][k, ph] == pytest.approx(beta, rel=5e-5, abs=2e-6)
                assert out["se"][k, ph] == pytest.approx(se, rel=5e-5)
                assert out["chisq"][k, ph] == pytest.approx(chisq, rel=1e-4, abs=2e-6)
                assert s2.get_logp(out["chisq"][k, ph]) == pytest.approx(logp, rel=1e-4, abs=2e-6)
        row += sel.size
    assert row == len(refs[0][1]) == 500
    assert nsparse > 100 and ndense > 100                      # both branches exercised
    assert moved > 1e-3                                        # and the sparse branch is not the dense number


def test_step2_bt_oracle_against_reference():
    """The binary-trait score test (no Firth / SPA): null logistic model with the LOCO offset per chromosome, compute_score_bt,
    get_sumstats -- oracle/regenie_step2_bt.py against regenie's own output for the documented Step-2 command without --firth
    (docs/docs/options.md: --step 2 --bed example --remove fid_iid_to_remove.txt --bt), 1,000 variants x 2 traits."""
    from oracle import regenie_step2_bt as bt
    from oracle import regenie_step2_qt as s2
    opt = orc.Step1Options(bed=E("example"), pheno_file=E("phenotype_bin.txt"), covar_file=E("covariates.txt"), bsize=200, bt=True,
                           remove=(E("fid_iid_to_remove.txt"),), test_mode=True)
    bim, chrom, offs, snp_ids, prep = orc.load_inputs(opt)
    bed, _ = orc.open_bed(opt.bed + ".bed", prep.n_file)
    ia = prep.ind_in_analysis
    ids = [i for i, k in zip(prep.ids, ia) if k]
    P = prep.Y.shape[1]
    loco = []
    for ph in range(P):
        hdr, v = read_loco_gz(os.path.join(REF_OUT, "bt_loocv_refcmd", "out_%d.loco.gz" % (ph + 1)))
        pos = {s: k for k, s in enumerate(hdr)}
        loco.append(v[:, [pos[i] for i in ids]])
    refs = [_read_regenie(os.path.join(REF_OUT, "step2", "bt_score_bed_Y%d.regenie.gz" % (ph + 1))) for ph in range(P)]
    col = {nm: i for i, nm in enumerate(refs[0][0])}
    X, Yraw, mask = prep.X[ia], prep.Y_raw[ia], prep.mask[ia]
    rows = {ph: {r[col["ID"]]: r for r in refs[ph][1]} for ph in range(P)}
    seen = 0
    for c in sorted(set(chrom.tolist())):
        nulls = [bt.null_logistic(Yraw[:, ph], X, mask[:, ph], loco[ph][c - 1], opt) for ph in range(P)]
        assert all(nl is not None for nl in nulls)
        sel = np.flatnonzero(chrom == c)
        G = orc.decode_bed_rows(np.asarray(bed[offs[sel]]), prep.n_file)[:, ~prep.ind_ignore][:, ia]
        for k in range(sel.size):
            g, mean, nobs = s2.mean_impute(G[k])
            for ph in range(P):
                r = rows[ph].get(snp_ids[sel[k]])
                if r is None:                      # dropped by the MAC filter
                    continue
                out = bt.score_bt(g, X, Yraw[:, ph], mask[:, ph].astype(np.float64), nulls[ph])
                beta, se, chisq, logp = (float(r[col[nm]]) for nm in ("BETA", "SE", "CHISQ", "LOG10P"))
                assert out["bhat"] == pytest.approx(beta, rel=5e-5, abs=2e-6)
                assert out["se"] == pytest.approx(se, rel=5e-5)
                assert out["chisq"] == pytest.approx(chisq, rel=1e-4, abs=2e-6)
                assert s2.get_logp(out["chisq"]) == pytest.approx(logp, rel=1e-4, abs=2e-6)
                seen += 1
    assert seen == sum(len(refs[ph][1]) for ph in range(P)) and seen > 1500


def test_step2_ct_oracle_against_reference(tmp_path):
    """The count-trait score test: null Poisson model with the LOCO offset per chromosome, compute_score_ct, get_sumstats --
    oracle/regenie_step2_bt.py against regenie's own --step 2 --ct output on synthetic counts (300 variants x 2 traits; the LOCO
    files are regenie's --qt predictions on the same file, see tests/golden/make_ref_outputs.py)."""
    import json
    from oracle import regenie_step2_bt as bt
    from oracle import regenie_step2_qt as s2
    from tests.util import synth_dosages, write_plink
    meta = json.load(open(os.path.join(REF_OUT, "ct_synth", "meta.json")))
    spec = meta["synthetic"]
    S = str(tmp_path / "synth")
    write_plink(S, synth_dosages(spec["M"], spec["N"], miss_rate=spec.get("miss_rate", 0.0), seed=spec["seed"]), spec["chroms"], P=spec["P"],
                seed=spec["seed"], binary=spec["binary"], missing_pheno=spec["missing_pheno"], counts=True)
    opt = orc.Step1Options(bed=S, pheno_file=S + ".pheno", covar_file=S + ".covar", bsize=100, ct=True, test_mode=True)
    bim, chrom, offs, snp_ids, prep = orc.load_inputs(opt)
    bed, _ = orc.open_bed(opt.bed + ".bed", prep.n_file)
    ia = prep.ind_in_analysis
    ids = [i for i, k in zip(prep.ids, ia) if k]
    P = prep.Y.shape[1]
    loco = []
    for ph in range(P):
        hdr, v = read_loco_gz(os.path.join(REF_OUT, "ct_synth", "out_%d.loco.gz" % (ph + 1)))
        pos = {s: k for k, s in enumerate(hdr)}
        loco.append(v[:, [pos[i] for i in ids]])
    refs = [_read_regenie(os.path.join(REF_OUT, "step2", "ct_synth_Y%d.regenie.gz" % (ph + 1))) for ph in range(P)]
    col = {nm: i for i, nm in enumerate(refs[0][0])}
    X, Yraw, mask = prep.X[ia], prep.Y_raw[ia], prep.mask[ia]
    rows = {ph: {r[col["ID"]]: r for r in refs[ph][1]} for ph in range(P)}
    seen = 0
    for c in sorted(set(chrom.tolist())):
        nulls = [bt.null_poisson(Yraw[:, ph], X, mask[:, ph], loco[ph][c - 1], opt) for ph in range(P)]
        assert all(nl is not None for nl in nulls)
        sel = np.flatnonzero(chrom == c)
        G = orc.decode_bed_rows(np.asarray(bed[offs[sel]]), prep.n_file)[:, ~prep.ind_ignore][:, ia]
        for k in range(sel.size):
            g, mean, nobs = s2.mean_impute(G[k])
            for ph in range(P):
                r = rows[ph].get(snp_ids[sel[k]])
                if r is None:
                    continue
                out = bt.score_ct(g, X, Yraw[:, ph], mask[:, ph].astype(np.float64), nulls[ph])
                beta, se, chisq, logp = (float(r[col[nm]]) for nm in ("BETA", "SE", "CHISQ", "LOG10P"))
                assert out["bhat"] == pytest.approx(beta, rel=5e-5, abs=2e-6)
                assert out["se"] == pytest.approx(se, rel=5e-5)
                assert out["chisq"] == pytest.approx(chisq, rel=1e-4, abs=2e-6)
                assert s2.get_logp(out["chisq"]) == pytest.approx(logp, rel=1e-4, abs=2e-6)
                seen += 1
    assert seen == sum(len(refs[ph][1]) for ph in range(P)) == 600


def test_step2_bt_approx_firth_oracle_against_reference_and_golden():
    """`--step 2 --bt --firth --approx --pThresh 0.01` on example.bgen, the command behind the ONE golden output the reference holds
    (example/test_bin_out_firth_Y1.regenie): null logistic + null Firth model per chromosome, score test, and for |z| above the
    threshold the 1-parameter Firth fit with the covariate effects in the offset (oracle/regenie_step2_bt.py, exact maximisers).
    Against the output of oracle/_ref/regenie for both traits (1,000 variants each, ~29 corrected rows) and against the golden
    file itself."""
    from oracle import bgen as obg
    from oracle import regenie_step2_bt as bt
    from oracle import regenie_step2_qt as s2
    opt = orc.Step1Options(bed=E("example"), pheno_file=E("phenotype_bin.txt"), covar_file=E("covariates.txt"), bsize=200, bt=True,
                           remove=(E("fid_iid_to_remove.txt"),), test_mode=True)
    bim, chrom, offs, snp_ids, prep = orc.load_inputs(opt)
    ia = prep.ind_in_analysis
    ids = [i for i, k in zip(prep.ids, ia) if k]
    P = prep.Y.shape[1]
    loco = []
    for ph in range(P):
        hdr, v = read_loco_gz(os.path.join(REF_OUT, "bt_loocv_refcmd", "out_%d.loco.gz" % (ph + 1)))
        pos = {s: k for k, s in enumerate(hdr)}
        loco.append(v[:, [pos[i] for i in ids]])
    refs = [_read_regenie(os.path.join(REF_OUT, "step2", "bt_firth_bgen_Y%d.regenie.gz" % (ph + 1))) for ph in range(P)]
    golden = _read_regenie(E("test_bin_out_firth_Y1.regenie"))
    exact = [_read_regenie(os.path.join(REF_OUT, "step2", "bt_firth_exact_bgen_Y%d.regenie.gz" % (ph + 1))) for ph in range(P)]
    erow = {ph: {r[2]: r for r in exact[ph][1]} for ph in range(P)}          # by variant ID
    spa = [_read_regenie(os.path.join(REF_OUT, "step2", "bt_spa_bgen_Y%d.regenie.gz" % (ph + 1))) for ph in range(P)]
    srow = {ph: {r[2]: r for r in spa[ph][1]} for ph in range(P)}
    col = {nm: i for i, nm in enumerate(refs[0][0])}
    X, Yraw, mask = prep.X[ia], prep.Y_raw[ia], prep.mask[ia]
    bg = obg.BgenOracle(E("example.bgen"))
    keep = ~prep.ind_ignore
    zthr = 2.5758293035489004                                   # sqrt of the 0.99 quantile of chi-square(1): --pThresh 0.01 (Data.cpp:2119-2120)
    rows = {ph: {r[col["ID"]]: r for r in refs[ph][1]} for ph in range(P)}
    grow = {r[col["ID"]]: r for r in golden[1]}
    nfirth = 0
    for c in sorted(set(chrom.tolist())):
        nulls, offs_f, bnulls = [], [], []
        for ph in range(P):
            nl = bt.null_logistic(Yraw[:, ph], X, mask[:, ph], loco[ph][c - 1], opt)
            bnull = bt.firth_null(Yraw[:, ph], X, mask[:, ph], loco[ph][c - 1], nl["beta"])
            assert nl is not None and bnull is not None
            nulls.append(nl)
            bnulls.append(bnull)
            offs_f.append(X @ bnull + loco[ph][c - 1])           # cov_blup_offset (Step2_Models.cpp:1011-1013)
        for k in np.flatnonzero(chrom == c):
            gk, flipped = bt.flip_geno(bg.dosages(int(k))[keep][ia])      # regenie tests the minor allele and negates BETA back (flip_geno, Geno.cpp:3150-3162)
            sgn = -1.0 if flipped else 1.0
            g, _, _ = s2.mean_impute(gk)
            for ph in range(P):
                r = rows[ph].get(snp_ids[k])
                if r is None:
                    continue
                m = mask[:, ph].astype(np.float64)
                out = bt.score_bt(g, X, Yraw[:, ph], m, nulls[ph])
                corrected = abs(out["stats"]) > zthr
                if corrected:
                    ex = bt.exact_firth(g, X, Yraw[:, ph], m, loco[ph][c - 1], bnulls[ph])       # --firth without --approx: same threshold, other fit
                    re = erow[ph][snp_ids[k]]
                    assert ex is not None
                    for nm in ("BETA", "SE", "CHISQ"):
                        assert ex[{"BETA": "bhat", "SE": "se", "CHISQ": "chisq"}[nm]] * (sgn if nm == "BETA" else 1.0) == pytest.approx(float(re[col[nm]]), rel=2e-4), (snp_ids[k], ph, nm)
                    sparse = s2.check_sparse(g, int(keep.sum()))                                   # fastSPA = the variant is sparse (Data.cpp:2503; Geno.cpp:3179)
                    sp = bt.spa_test(out["stats"], out["denum"], out["Gres"], nulls[ph], m, carriers=np.flatnonzero(g != 0) if sparse else None)   # --spa
                    rs = srow[ph][snp_ids[k]]
                    assert sp is not None
                    for nm, key in (("BETA", "bhat"), ("SE", "se"), ("CHISQ", "chisq"), ("LOG10P", "logp")):
                        assert sp[key] * (sgn if key == "bhat" else 1.0) == pytest.approx(float(rs[col[nm]]), rel=3e-5), (snp_ids[k], ph, nm)
                    out = bt.approx_firth(g, X, Yraw[:, ph], m, nulls[ph], offs_f[ph])
                    assert out is not None
                    nfirth += 1
                for rr, tol in ((r, 5e-5),) + (((grow[snp_ids[k]], 2e-4),) if ph == 0 else ()):
                    beta, se, chisq, logp = (float(rr[col[nm]]) for nm in ("BETA", "SE", "CHISQ", "LOG10P"))
                    assert sgn * out["bhat"] == pytest.approx(beta, rel=tol, abs=2e-6), (snp_ids[k], ph, corrected)
                    assert out["se"] == pytest.approx(se, rel=tol)
                    assert out["chisq"] == pytest.approx(chisq, rel=2 * tol, abs=2e-6)
                    assert s2.get_logp(out["chisq"]) == pytest.approx(logp, rel=2 * tol, abs=2e-6)
    assert nfirth >= 25


def test_step2_bt_approx_firth_rare_variants_against_reference(tmp_path):
    """Rare, sparse variants (MAF 0.1 - 1 %, 5,200 samples, 136 of 300 variants with MAC < 50): regenie's approximate Firth fit then keeps the
    carriers only (fit_firth_logistic_snp_fast, Step2_Models.cpp:1173-1185).  --pThresh 0.3, so ~270 of the 900 tests are corrected.  Oracle
    against regenie's own output (tests/golden/ref_outputs/step2/bt_firth_rare_Y*.regenie.gz; LOCO files of the bt_kfold_synth case)."""
    import json
    from scipy.stats import norm
    from oracle import regenie_step2_bt as bt
    from oracle import regenie_step2_qt as s2
    from tests.util import synth_dosages, synth_rare_dosages, write_bed_bim, write_plink
    meta = json.load(open(os.path.join(REF_OUT, "bt_kfold_synth", "meta.json")))
    spec = meta["synthetic"]
    S = str(tmp_path / "synth")
    write_plink(S, synth_dosages(spec["M"], spec["N"], miss_rate=spec.get("miss_rate", 0.0), seed=spec["seed"]), spec["chroms"], P=spec["P"],
                seed=spec["seed"], binary=spec["binary"], missing_pheno=spec["missing_pheno"])
    write_bed_bim(S + "_rare", synth_rare_dosages(300, spec["N"], seed=spec["seed"], miss_rate=0.002), [1] * 100 + [2] * 100 + [5] * 100)
    shutil.copy(S + ".fam", S + "_rare.fam")
    opt = orc.Step1Options(bed=S + "_rare", pheno_file=S + ".pheno", covar_file=S + ".covar", bsize=100, bt=True, test_mode=True)
    bim, chrom, offs, snp_ids, prep = orc.load_inputs(opt)
    bed, _ = orc.open_bed(opt.bed + ".bed", prep.n_file)
    ia = prep.ind_in_analysis
    ids = [i for i, k in zip(prep.ids, ia) if k]
    P = prep.Y.shape[1]
    loco = []
    for ph in range(P):
        hdr, v = read_loco_gz(os.path.join(REF_OUT, "bt_kfold_synth", "out_%d.loco.gz" % (ph + 1)))
        pos = {s: k for k, s in enumerate(hdr)}
        loco.append(v[:, [pos[i] for i in ids]])
    refs = [_read_regenie(os.path.join(REF_OUT, "step2", "bt_firth_rare_Y%d.regenie.gz" % (ph + 1))) for ph in range(P)]
    spa = [_read_regenie(os.path.join(REF_OUT, "step2", "bt_spa_rare_Y%d.regenie.gz" % (ph + 1))) for ph in range(P)]
    srow = {ph: {r[2]: r for r in spa[ph][1]} for ph in range(P)}
    col = {nm: i for i, nm in enumerate(refs[0][0])}
    X, Yraw, mask = prep.X[ia], prep.Y_raw[ia], prep.mask[ia]
    zthr = float(norm.ppf(1 - 0.3 / 2))
    nspa_fail = 0
    n_all = int((~prep.ind_ignore).sum())
    ncorr = nfast = 0
    for c in sorted(set(chrom.tolist())):
        nulls, offs_f = [], []
        for ph in range(P):
            nl = bt.null_logistic(Yraw[:, ph], X, mask[:, ph], loco[ph][c - 1], opt)
            bnull = bt.firth_null(Yraw[:, ph], X, mask[:, ph], loco[ph][c - 1], nl["beta"])
            nulls.append(nl)
            offs_f.append(X @ bnull + np.nan_to_num(loco[ph][c - 1]))      # NA predictions belong to samples masked for the trait
        sel = np.flatnonzero(chrom == c)
        G = orc.decode_bed_rows(np.asarray(bed[offs[sel]]), prep.n_file)[:, ~prep.ind_ignore][:, ia]
        rows = {ph: {r[col["ID"]]: r for r in refs[ph][1]} for ph in range(P)}
        for k in range(sel.size):
            gk, flipped = bt.flip_geno(G[k])                      # the minor allele is what regenie tests (flip_geno), BETA negated back
            sgn = -1.0 if flipped else 1.0
            g, _, _ = s2.mean_impute(gk)
            sparse = s2.check_sparse(g, n_all)
            obs = G[k] >= 0
            for ph in range(P):
                r = rows[ph].get(snp_ids[sel[k]])
                if r is None:
                    continue
                m = mask[:, ph].astype(np.float64)
                out = bt.score_bt(g, X, Yraw[:, ph], m, nulls[ph], sparse=sparse)
                if abs(out["stats"]) > zthr:
                    # --spa on the same data (bt_spa_rare): the fast form for sparse variants, one test regenie reports as TEST_FAIL
                    sp = bt.spa_test(out["stats"], out["denum"], out["Gres"], nulls[ph], m, carriers=np.flatnonzero(g != 0) if sparse else None)
                    rs = srow[ph][snp_ids[sel[k]]]
                    if rs[-1] == "TEST_FAIL":
                        assert sp is None and sgn * out["bhat"] == pytest.approx(float(rs[col["BETA"]]), rel=3e-5)
                        nspa_fail += 1
                    else:
                        for nm, key in (("BETA", "bhat"), ("SE", "se"), ("CHISQ", "chisq"), ("LOG10P", "logp")):
                            assert sp[key] * (sgn if key == "bhat" else 1.0) == pytest.approx(float(rs[col[nm]]), rel=3e-5), (snp_ids[sel[k]], ph, nm)
                    tq = float(G[k][obs & (m > 0)].sum())
                    nq = int((obs & (m > 0)).sum())
                    mac = min(tq, 2 * nq - tq)
                    out = bt.approx_firth(g, X, Yraw[:, ph], m, nulls[ph], offs_f[ph], sparse=sparse, mac=mac)
                    assert out is not None
                    ncorr += 1
                    nfast += sparse and mac < 50
                beta, se, chisq = (float(r[col[nm]]) for nm in ("BETA", "SE", "CHISQ"))
                # regenie stops its 1-parameter fit at |modified score| < 2.5e-4, i.e. within a few times 2.5e-4 * se^2 of the root this oracle finds
                assert abs(sgn * out["bhat"] - beta) <= 8e-4 * se * se + 2e-5 * abs(beta) + 5e-6, (snp_ids[sel[k]], ph)
                assert out["se"] == pytest.approx(se, rel=2e-4)
                assert out["chisq"] == pytest.approx(chisq, rel=2e-3, abs=2e-5)      # (its LRT is taken one outer iteration before its BETA)
    assert ncorr > 200 and nfast > 80 and nspa_fail == 1


needs_ref_binary = pytest.mark.skipif(not os.path.exists(REGENIE), reason="oracle/_ref/regenie not built (make -C oracle)")


@needs_ref_binary
def test_reference_build_reproduces_the_reference_held_golden(tmp_path):
    """docs/docs/options.md:20-53: the two documented commands; example/test_bin_out_firth_Y1.regenie is the reference's
    own copy of the output.  Rows without the Firth correction (closed-form score test) must match byte for byte; the
    Firth rows (iterative fit, tolerance-stopped) to 1e-4."""
    d = str(tmp_path)
    os.symlink(E("example.bgen"), os.path.join(d, "ex.bgen"))
    base = ["--covarFile", E("covariates.txt"), "--phenoFile", E("phenotype_bin.txt"), "--remove", E("fid_iid_to_remove.txt"), "--bt"]
    r = subprocess.run([REGENIE, "--step", "1", "--bed", E("example"), "--exclude", E("snplist_rm.txt"), "--bsize", "100",
                        "--lowmem", "--lowmem-prefix", "tmp_rg", "--out", "fit_bin_out"] + base, cwd=d, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert any("0.4504" in ln and "min value" in ln for ln in r.stdout.splitlines())     # test/test_bash.sh:87
    r = subprocess.run([REGENIE, "--step", "2", "--bgen", "ex.bgen", "--bsize", "200", "--firth", "--approx", "--pThresh", "0.01",
                        "--pred", "fit_bin_out_pred.list", "--out", "test_bin_out_firth"] + base, cwd=d, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    got = open(os.path.join(d, "test_bin_out_firth_Y1.regenie")).read().splitlines()
    ref = open(E("test_bin_out_firth_Y1.regenie")).read().splitlines()
    assert len(got) == len(ref) == 1001 and got[0] == ref[0]
    same = sum(1 for a, b in zip(got, ref) if a == b)
    assert same >= 975, same
    for a, b in zip(got, ref):
        if a != b:
            ta, tb = a.split(), b.split()
            assert ta[:9] == tb[:9]
            for x, y in zip(ta[9:13], tb[9:13]):
                assert float(x) == pytest.approx(float(y), rel=1e-4)
    # the committed Step-1 fixture of the same command is what this binary writes
    for ph in (1, 2):
        assert open(os.path.join(d, "fit_bin_out_%d.loco" % ph), "rb").read() == \
            gzip.open(os.path.join(REF_OUT, "bt_loocv_refcmd", "out_%d.loco.gz" % ph), "rb").read()


@pytest.mark.skipif(not os.path.exists(REGENIE), reason="oracle/_ref/regenie not built")
def test_oracle_against_live_reference_on_drawn_cases(tmp_path):
    """A few cases of tests/golden/fuzz_oracle_vs_reference.py (routes, sizes, block size, folds, grid sizes, --ref-first / --strict, missing
    genotypes / phenotypes drawn per seed): regenie runs here and the oracle is held to its .loco files, tables and Step-2 statistics.  The
    400-case run of the round is tests/golden/fuzz_log.md."""
    from tests.golden import fuzz_oracle_vs_reference as fz
    for seed in (1, 2, 3, 4, 6):          # QT leave-one-out, BT leave-one-out, QT K-fold, BT K-fold, --ref-first
        line, ok = fz.run_one(seed, str(tmp_path))
        assert ok, line


def test_null_firth_estimates_against_reference_files(tmp_path):
    """--write-null-firth of regenie on the bt_kfold_synth case (tests/golden/ref_outputs/bt_kfold_synth/out_k.firth.gz: per chromosome the covariate
    estimates of the null Firth model with that chromosome's LOCO prediction as offset; three traits with 100 - 121 of 5,200 samples masked): the
    oracle's firth_null within 1e-4 -- which it is only with regenie's treatment of the masked samples (weight 1 in X^T W X: get_wvec,
    Step1_Models.cpp:1809-1811; fit_firth_nr, Step2_Models.cpp:1287-1290); leaving their rows out is 9e-4 away on the second trait."""
    import json
    from oracle import regenie_step2_bt as bt
    from tests.util import synth_dosages, write_plink
    meta = json.load(open(os.path.join(REF_OUT, "bt_kfold_synth", "meta.json")))
    spec = meta["synthetic"]
    S = str(tmp_path / "synth")
    write_plink(S, synth_dosages(spec["M"], spec["N"], miss_rate=spec.get("miss_rate", 0.0), seed=spec["seed"]), spec["chroms"], P=spec["P"],
                seed=spec["seed"], binary=spec["binary"], missing_pheno=spec["missing_pheno"])
    opt = orc.Step1Options(bed=S, pheno_file=S + ".pheno", covar_file=S + ".covar", bsize=100, bt=True, test_mode=True)
    bim, chrom, offs, snp_ids, prep = orc.load_inputs(opt)
    ia = prep.ind_in_analysis
    ids = [i for i, k in zip(prep.ids, ia) if k]
    X, Yraw, mask = prep.X[ia], prep.Y_raw[ia], prep.mask[ia]
    without = 0.0
    for ph in range(prep.Y.shape[1]):
        hdr, v = read_loco_gz(os.path.join(REF_OUT, "bt_kfold_synth", "out_%d.loco.gz" % (ph + 1)))
        pos = {s: k for k, s in enumerate(hdr)}
        loco = np.nan_to_num(v[:, [pos[i] for i in ids]])
        rows = [ln.split() for ln in gzip.open(os.path.join(REF_OUT, "bt_kfold_synth", "out_%d.firth.gz" % (ph + 1)), "rt").read().splitlines()]
        m = mask[:, ph]
        assert (~m).sum() >= 100
        for row in rows[:2]:
            c, want = int(row[0]), np.array([float(t) for t in row[1:]])
            nl = bt.null_logistic(Yraw[:, ph], X, m, loco[c - 1], opt)
            got = bt.firth_null(Yraw[:, ph], X, m, loco[c - 1], nl["beta"])
            sign = np.sign(got * want)                                  # the basis columns are defined up to sign
            assert np.max(np.abs(got * sign - want) / np.abs(want)) < 1e-4, (ph, c)
            alone = bt.firth_null(Yraw[m, ph], X[m], np.ones(int(m.sum()), bool), loco[c - 1][m], nl["beta"])
            without = max(without, float(np.max(np.abs(alone * np.sign(alone * want) - want) / np.abs(want))))
    assert without > 5e-4
