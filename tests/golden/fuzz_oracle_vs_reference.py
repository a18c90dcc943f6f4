#!/usr/bin/env python
"""Randomised pinning of the Step-1 oracle against regenie ITSELF (CPU only; needs oracle/_ref/regenie, i.e. this container).

The committed fixtures (make_ref_outputs.py) pin the oracle on a fixed set of cases.  This script draws cases -- route (QT K-fold, QT
leave-one-out, BT leave-one-out, BT K-fold at >= 5,000 samples), sample / variant / chromosome / phenotype counts, block size, folds,
ridge-grid sizes, --ref-first, --strict, missing genotypes and phenotypes -- on the synthetic data of tests/util.py, runs regenie v4.1.2 on
each, and holds oracle/regenie_step1.py to its .loco files (at the text's resolution), CV tables and selected ridge values with the
assertions of tests/test_reference_pin.py.  On the quantitative-trait cases regenie's --step 2 --qt then runs on the same files with its own
LOCO predictions, and oracle/regenie_step2_qt.py (score_qt_block_ref: the sparse / dense choice per variant) is held to every BETA / SE /
CHISQ / LOG10P of its .regenie files.  Usage:  python tests/golden/fuzz_oracle_vs_reference.py [first_seed=1] [count=40] [log.md]
A line per case goes to stdout (and to the log file); a mismatch is printed with its arguments and the script exits 1 at the end.
FUZZ_DRIVER=1 (GPU box): the PRODUCT runs beside them -- `regenie-amd --step 1` and `--step 2 --qt` with the same arguments -- and its .loco
files and .regenie lines are held to regenie's (values at the text's resolution; the share of byte-identical lines is reported); with FUZZ_BGEN /
FUZZ_PGEN also on those inputs (driver_same: written in round 5 after the GPU budget was spent -- exercised with regenie standing in for the
driver, first real run due in the next round:  FUZZ_DRIVER=1 FUZZ_BGEN=2 FUZZ_PGEN=1 FUZZ_PREP=2 python tests/golden/fuzz_oracle_vs_reference.py 1 100).
Other switches: FUZZ_PREP=1|2 (host-preparation options), FUZZ_ROUTES=ct_kfold,ct_loocv,t2e_kfold,... (routes to cycle through), FUZZ_BT_STEP2=1|2|3
(the binary score test / its Firth and saddlepoint corrections), FUZZ_BGEN=1|2, FUZZ_PGEN=1.
FUZZ_BUDGET_S=t stops drawing new cases after t seconds."""
import os
import subprocess
import sys
import tempfile
import time
import traceback

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import regenie_step1 as orc                                   # noqa: E402
from tests import test_reference_pin as pin                              # noqa: E402
from tests.golden.make_ref_outputs import table_lines                    # noqa: E402
from tests.util import synth_dosages, write_plink                        # noqa: E402

REGENIE = os.path.join(ROOT, "oracle", "_ref", "regenie")
BIN = os.environ.get("FUZZ_DRIVER_BIN") or os.path.join(ROOT, "regenie_amd", "bin", "regenie-amd")
# FUZZ_DRIVER_BIN=<tests/hipcpu/emubuild.py's regenie-amd-hostbt> FUZZ_DRIVER_BT_ONLY=1: the driver with the binary-trait Step-2 kernels running on the
# host stand-in of the HIP runtime (no GPU): only the corrected-rows leg (FUZZ_BT_STEP2=2) has a driver run then


def _loco(path):
    lines = open(path).read().splitlines()
    return lines[0].split()[1:], np.array([[np.nan if t == "NA" else float(t) for t in ln.split()[1:]] for ln in lines[1:]])


def driver_same(d, args, ref, kind):
    """FUZZ_DRIVER on another input format: `regenie-amd` with the arguments regenie just ran with (outputs `ref`_*), its files held to regenie's --
    .loco files within 1e-5 of the largest value (kind "step1"), .regenie lines equal or within 2e-5 per number (kind "step2").  -> text"""
    if not os.environ.get("FUZZ_DRIVER"):
        return ""
    drv = "d" + ref
    r = subprocess.run([BIN] + args + ["--out", drv], cwd=d, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, "regenie-amd (%s): " % ref + (r.stdout + r.stderr)[-500:]
    names = [ln.split()[0] for ln in open(os.path.join(d, ref + "_pred.list"))] if kind == "step1" else None
    if kind == "step1":
        assert [ln.split()[0] for ln in open(os.path.join(d, drv + "_pred.list"))] == names
        same = 0
        listed = [os.path.basename(ln.split()[1]) for ln in open(os.path.join(d, ref + "_pred.list"))]
        for fn in listed:
            ids_r, a = _loco(os.path.join(d, fn))
            ids_g, b = _loco(os.path.join(d, "d" + fn))
            assert ids_r == ids_g and np.array_equal(np.isnan(a), np.isnan(b))
            ok = ~np.isnan(a)
            assert float(np.max(np.abs(a[ok] - b[ok])) / np.max(np.abs(a[ok]))) < 1e-5, "driver .loco (%s)" % ref
            same += open(os.path.join(d, fn)).read() == open(os.path.join(d, "d" + fn)).read()
        return " [driver: %d/%d files byte-identical]" % (same, len(listed))
    same = tot = 0
    for fn in sorted(f for f in os.listdir(d) if f.startswith(ref + "_") and f.endswith(".regenie")):
        a = open(os.path.join(d, "d" + fn)).read().splitlines()
        b = open(os.path.join(d, fn)).read().splitlines()
        assert a[0] == b[0] and len(a) == len(b), "driver .regenie header / line count (%s)" % ref
        for x, y in zip(a[1:], b[1:]):
            tot += 1
            if x == y:
                same += 1
                continue
            tx, ty = x.split(" "), y.split(" ")
            assert len(tx) == len(ty) and tx[:5] == ty[:5], (x, y)
            for u, v in zip(tx[5:], ty[5:]):
                assert u == v or (u != "NA" and v != "NA" and abs(float(u) - float(v)) <= 2e-5 * abs(float(v)) + 2e-9), (x, y)
    return " [driver: %d/%d lines byte-identical]" % (same, tot)


def driver_legs(d, args1, o, P):
    """The product on the case: its .loco files against regenie's (1e-5 of the largest value = BASELINE.json's bar; text ulps reported), and for
    quantitative traits its --step 2 --qt lines against regenie's."""
    r = subprocess.run([BIN] + args1 + ["--out", "drv"], cwd=d, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, "regenie-amd --step 1: " + (r.stdout + r.stderr)[-500:]
    assert open(os.path.join(d, "drv_pred.list")).read().replace("drv_", "out_") == open(os.path.join(d, "out_pred.list")).read()
    skipped = lambda t: [ln.split(" : ")[0] for ln in t.splitlines() if ln.startswith("phenotype ") and "did not converge" in ln]      # noqa: E731
    assert skipped(open(os.path.join(d, "drv.log")).read()) == skipped(open(os.path.join(d, "out.log")).read()), "traits reported as not converged"
    worst_ulp, worst_rel, same_files = 0.0, 0.0, 0
    for ph in range(P):
        ids_r, ref = _loco(os.path.join(d, "out_%d.loco" % (ph + 1)))
        ids_g, got = _loco(os.path.join(d, "drv_%d.loco" % (ph + 1)))
        assert ids_r == ids_g and got.shape == ref.shape and np.array_equal(np.isnan(got), np.isnan(ref)), "driver .loco layout / NA pattern"
        ok = ~np.isnan(ref)
        mag = np.maximum(np.abs(ref[ok]), 1e-300)
        ulp = 10.0 ** (np.floor(np.log10(mag)) - 5)
        err = np.abs(got[ok] - ref[ok])
        worst_ulp = max(worst_ulp, float(np.max(err / ulp)))
        worst_rel = max(worst_rel, float(np.max(err) / np.max(np.abs(ref[ok]))))
        same_files += open(os.path.join(d, "out_%d.loco" % (ph + 1))).read() == open(os.path.join(d, "drv_%d.loco" % (ph + 1))).read()
    assert worst_rel < 1e-5, "driver .loco: %.2e of the largest value" % worst_rel
    out = "driver: loco %.1e (%.1f text ulps, %d/%d files byte-identical)" % (worst_rel, worst_ulp, same_files, P)
    if not o["bt"] and not o.get("ct"):
        S = os.path.join(d, "synth")
        a2 = ["--step", "2", "--qt", "--bed", S, "--phenoFile", S + ".pheno", "--covarFile", S + ".covar", "--bsize", "200", "--pred", "out_pred.list"]
        a2 += ["--ref-first"] if o["ref_first"] else []
        a2 += ["--strict"] if o["strict"] else []
        a2 += _prep_args(S, o)
        r = subprocess.run([BIN] + a2 + ["--out", "d2"], cwd=d, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, "regenie-amd --step 2: " + (r.stdout + r.stderr)[-500:]
        same = tot = 0
        for nm in [ln.split()[0] for ln in open(os.path.join(d, "out_pred.list"))]:
            a = open(os.path.join(d, "d2_%s.regenie" % nm)).read().splitlines()
            b = open(os.path.join(d, "s2_%s.regenie" % nm)).read().splitlines()
            assert a[0] == b[0] and len(a) == len(b), "driver .regenie header / line count"
            for x, y in zip(a[1:], b[1:]):
                tot += 1
                if x == y:
                    same += 1
                    continue
                tx, ty = x.split(" "), y.split(" ")
                assert len(tx) == len(ty) and tx[:5] == ty[:5], (x, y)
                for u, v in zip(tx[5:], ty[5:]):
                    assert u == v or (u != "NA" and v != "NA" and abs(float(u) - float(v)) <= 2e-5 * abs(float(v)) + 2e-9), (x, y)
        out += ", step 2 %d/%d lines byte-identical" % (same, tot)
    return out



def ref_table(log_text):
    """table_lines() that also keeps a trait regenie skipped: `phenotype k (name) : Level 1 model did not converge. ...` is ONE line of its log"""
    out = []
    for ln in log_text.splitlines():
        if ln.startswith("phenotype ") and " : " in ln + " " and "(" in ln:
            out.append(ln.split(" : ")[0].rstrip().rstrip(":").rstrip() + " :" if "did not converge" in ln else ln.rstrip())
        elif ": Rsq = " in ln:
            out.append(ln.rstrip())
    return out


def draw(seed):
    rng = np.random.default_rng(100000 + seed)
    cycle = os.environ["FUZZ_ROUTES"].split(",") if os.environ.get("FUZZ_ROUTES") else ["qt_kfold", "qt_loocv", "bt_loocv", "qt_kfold", "bt_kfold"]
    route = cycle[seed % len(cycle)]          # FUZZ_ROUTES: also ct_kfold (count traits) and t2e_kfold (time-to-event traits, Cox ridge at level 1)
    if route == "bt_kfold":
        N, M = int(rng.integers(5050, 5400)), int(rng.integers(60, 140))
    else:
        N, M = int(rng.integers(250, 1100)), int(rng.integers(120, 420))
    nchr = int(rng.integers(2, 5))
    cuts = np.sort(rng.choice(np.arange(10, M - 10), nchr - 1, replace=False))
    chrom_ids = np.sort(rng.choice(np.arange(1, 23), nchr, replace=False))
    chroms = np.repeat(chrom_ids, np.diff(np.concatenate([[0], cuts, [M]]))).tolist()
    spec = {"M": M, "N": N, "chroms": chroms, "P": int(rng.integers(1, 4)), "seed": int(1000 + seed), "binary": route.startswith("bt"),
            "missing_pheno": float(rng.choice([0.0, 0.05])), "miss_rate": float(rng.choice([0.0, 0.02]))}
    opt = {"bsize": int(rng.choice([37, 64, 100, 150])), "bt": route.startswith("bt"), "loocv": route == "qt_loocv",
           "cv_folds": int(rng.choice([3, 4, 5, 7])), "n_ridge_l0": int(rng.choice([3, 5, 6])), "n_ridge_l1": int(rng.choice([4, 5, 7])),
           "ref_first": bool(rng.random() < 0.3), "strict": bool(spec["missing_pheno"] > 0 and rng.random() < 0.3)}
    if route in ("ct_kfold", "ct_loocv"):
        spec["counts"] = "poisson"
        opt["ct"] = True
        opt["loocv"] = route == "ct_loocv"
    if route == "t2e_kfold":
        spec["t2e"] = {"ntraits": spec["P"], "missing": spec["missing_pheno"] * 0.6, "decimals": int(rng.choice([1, 2, 3]))}
        spec["missing_pheno"] = 0.0
        opt["strict"] = False
    return route, spec, opt


def draw_prep(seed, route):
    """Host-preparation options on top of a case (FUZZ_PREP=1): --remove, --exclude, --apply-rint (quantitative traits), a categorical covariate."""
    rng = np.random.default_rng(700000 + seed)
    pr = {"remove": bool(rng.random() < 0.4), "exclude": bool(rng.random() < 0.4), "rint": bool(route.startswith("qt") and rng.random() < 0.4),
          "cat": bool(rng.random() < 0.4), "levels": int(rng.integers(2, 6)), "seed": int(rng.integers(1, 1 << 30))}
    if os.environ.get("FUZZ_PREP") == "2":      # the complementary lists, a subset of the phenotype columns, --nb
        pr.update(keep=bool(rng.random() < 0.4), extract=bool(rng.random() < 0.4), phenocol=bool(rng.random() < 0.4), nb=bool(rng.random() < 0.3))
        pr["keep"] = pr["keep"] and not pr["remove"]          # (regenie takes one of --keep / --remove)
    if os.environ.get("FUZZ_PREP") == "4":      # a subset of the covariate columns, 1 / 2 coding of a binary trait, --minCaseCount, few Newton rounds
        pr.update(covarcol=bool(rng.random() < 0.5), cc12=bool(route.startswith("bt") and rng.random() < 0.5), mincase=bool(route.startswith("bt") and rng.random() < 0.4),
                  niter=bool(route.startswith("bt") and rng.random() < 0.3))
    if os.environ.get("FUZZ_PREP") == "3":      # explicit ridge grids, leave-one-out forced on a binary trait, --print-prs
        pr.update(setl0=bool(rng.random() < 0.5), setl1=bool(rng.random() < 0.5), force_loocv=bool(route.startswith("bt") and rng.random() < 0.5), prs=bool(rng.random() < 0.4))
    return pr


def apply_prep(S, spec, pr):
    """Writes the files the options name; -> (regenie arguments, oracle options)"""
    rng = np.random.default_rng(pr["seed"])
    args, kw = [], {}
    if pr["remove"]:
        ids = np.sort(rng.choice(spec["N"], max(1, spec["N"] // 20), replace=False)) + 1
        open(S + ".remove", "w").write("".join("%d %d\n" % (i, i) for i in ids))
        args += ["--remove", S + ".remove"]; kw["remove"] = [S + ".remove"]
    if pr["exclude"]:
        ids = np.sort(rng.choice(spec["M"], max(1, spec["M"] // 15), replace=False))
        open(S + ".exclude", "w").write("".join("s%d\n" % i for i in ids))
        args += ["--exclude", S + ".exclude"]; kw["exclude"] = [S + ".exclude"]
    if pr.get("keep"):
        ids = np.sort(rng.choice(spec["N"], spec["N"] - max(1, spec["N"] // 12), replace=False)) + 1
        open(S + ".keep", "w").write("".join("%d %d\n" % (i, i) for i in ids))
        args += ["--keep", S + ".keep"]; kw["keep"] = [S + ".keep"]
    if pr.get("extract"):
        ids = np.sort(rng.choice(spec["M"], spec["M"] - max(1, spec["M"] // 10), replace=False))
        open(S + ".extract", "w").write("".join("s%d\n" % i for i in ids))
        args += ["--extract", S + ".extract"]; kw["extract"] = [S + ".extract"]
    if pr.get("phenocol") and spec["P"] > 1:
        cols = ["Y%d" % (k + 1) for k in np.sort(rng.choice(spec["P"], spec["P"] - 1, replace=False))]
        args += ["--phenoColList", ",".join(cols)]; kw["pheno_cols"] = cols
    if pr.get("nb"):
        nb = int(rng.integers(2, 5))
        args += ["--nb", str(nb)]; kw["n_block"] = nb
    if pr.get("covarcol"):
        args += ["--covarColList", "C2"]; kw["covar_cols"] = ["C2"]
    if pr.get("cc12"):             # control = 1, case = 2 (Pheno.cpp:260-270)
        lines = open(S + ".pheno").read().splitlines()
        out = [lines[0]] + [" ".join(t[:2] + [x if x == "NA" else str(int(float(x)) + 1) for x in t[2:]]) for t in (ln.split() for ln in lines[1:])]
        open(S + ".pheno", "w").write("\n".join(out) + "\n")
        args += ["--cc12"]; kw["cc12"] = True
    if pr.get("mincase"):
        mc = int(rng.choice([50, 200, 400]))
        args += ["--minCaseCount", str(mc)]; kw["min_case_count"] = mc
    if pr.get("niter"):
        ni = int(rng.choice([2, 4, 8]))
        args += ["--niter", str(ni)]; kw["niter_max"] = ni; kw["niter_max_ridge"] = ni            # (Regenie.cpp:483: both limits)
    if pr.get("setl0"):
        v = np.sort(rng.uniform(0.02, 0.95, int(rng.integers(2, 6))))
        args += ["--setl0", ",".join("%.4f" % x for x in v)]; kw["setl0"] = [float("%.4f" % x) for x in v]
    if pr.get("setl1"):
        v = np.sort(rng.uniform(0.02, 0.95, int(rng.integers(2, 6))))
        args += ["--setl1", ",".join("%.4f" % x for x in v)]; kw["setl1"] = [float("%.4f" % x) for x in v]
    if pr.get("force_loocv"):
        args += ["--loocv"]; kw["loocv"] = True
    if pr.get("prs"):
        args += ["--print-prs"]; kw["print_prs"] = True
    if pr["rint"]:
        args += ["--apply-rint"]; kw["apply_rint"] = True
    if pr["cat"]:
        lines = open(S + ".covar").read().splitlines()
        lev = rng.integers(0, pr["levels"], spec["N"])
        lev[: pr["levels"]] = rng.permutation(pr["levels"])            # every level occurs; the order of first appearance is drawn
        out = [lines[0] + " CAT"] + [ln + " %d" % (7 * int(v) + 3) for ln, v in zip(lines[1:], lev)]
        open(S + ".covar", "w").write("\n".join(out) + "\n")
        args += ["--catCovarList", "CAT"]; kw["cat_covar"] = ["CAT"]
    return args, kw


def run_one(seed, work):
    route, spec, o = draw(seed)
    d = os.path.join(work, "c%d" % seed)
    os.makedirs(d)
    S = os.path.join(d, "synth")
    g = synth_dosages(spec["M"], spec["N"], miss_rate=spec["miss_rate"], seed=spec["seed"])
    write_plink(S, g, spec["chroms"], P=spec["P"], seed=spec["seed"], binary=spec["binary"], missing_pheno=spec["missing_pheno"], counts=spec.get("counts", False))
    if route == "t2e_kfold":
        return run_t2e(seed, d, S, g, spec, o)
    args = ["--step", "1", "--bed", S, "--phenoFile", S + ".pheno", "--covarFile", S + ".covar", "--bsize", str(o["bsize"]), "--cv", str(o["cv_folds"]),
            "--l0", str(o["n_ridge_l0"]), "--l1", str(o["n_ridge_l1"])]
    args += ["--bt"] if o["bt"] else []
    args += ["--ct"] if o.get("ct") else []
    args += ["--loocv"] if o["loocv"] else []
    args += ["--ref-first"] if o["ref_first"] else []
    args += ["--strict"] if o["strict"] else []
    args += ["--write-null-firth"] if o["bt"] and os.environ.get("FUZZ_BT_NULLFIRTH") else []       # out_<k>.firth + out_firth.list for --use-null-firth in Step 2
    prep_desc = ""
    if os.environ.get("FUZZ_PREP"):
        pr = draw_prep(seed, route)
        pa, pk = apply_prep(S, spec, pr)
        args += pa
        o = dict(o, **pk)
        prep_desc = "".join(" " + k for k in ("remove", "exclude", "rint", "keep", "extract", "phenocol", "nb", "setl0", "setl1", "force_loocv", "prs", "covarcol", "cc12", "mincase", "niter") if pr.get(k)) + (" cat%d" % pr["levels"] if pr["cat"] else "")
    t0 = time.time()
    r = subprocess.run([REGENIE] + args + ["--out", "out"], cwd=d, capture_output=True, text=True)
    t_ref = time.time() - t0
    desc = "seed %d %-8s N %d M %d chr %d P %d bsize %d cv %d l0 %d l1 %d%s%s missG %.2f missY %.2f" % (
        seed, route, spec["N"], spec["M"], len(set(spec["chroms"])), spec["P"], o["bsize"], o["cv_folds"], o["n_ridge_l0"], o["n_ridge_l1"],
        " ref-first" if o["ref_first"] else "", " strict" if o["strict"] else "", spec["miss_rate"], spec["missing_pheno"]) + prep_desc
    if r.returncode != 0:
        return desc + " | regenie itself stopped: " + (r.stdout + r.stderr).strip().splitlines()[-1][:160], None
    log = open(os.path.join(d, "out.log")).read()
    t0 = time.time()
    res = orc.run_step1(orc.Step1Options(bed=S, pheno_file=S + ".pheno", covar_file=S + ".covar", **o))
    t_or = time.time() - t0
    ref_tab = pin.parse_table(ref_table(log))
    got_tab = pin.parse_table([l for l in res.log if l.startswith("phenotype ") or ": Rsq = " in l])
    names = [ln.split()[0] for ln in open(os.path.join(d, "out_pred.list"))]
    # a trait whose level-1 model did not converge has a line in the table ("... LOCO predictions calculations are skipped") and no file
    assert len(ref_tab) == len(got_tab), (len(ref_tab), len(got_tab))
    assert names == [res.prep.pheno_names[ph] for ph in range(len(got_tab)) if res.loco[ph] is not None], ("traits with predictions", names)
    skipped = len(got_tab) - len(names)
    worst = 0.0
    for ph, (rt, gt) in enumerate(zip(ref_tab, got_tab)):
        assert len(rt) == len(gt)
        for (h, rsq, mse, ll, mn), (h2, rsq2, mse2, ll2, mn2) in zip(rt, gt):
            assert h == h2 and mn == mn2, ("selected ridge value", ph, h)
            assert abs(rsq2 - rsq) <= 2e-5 * max(abs(rsq), 1e-12) + 1e-12 and (np.isnan(mse) or abs(mse2 - mse) <= 2e-5 * abs(mse)), ("table", ph, h, rsq, rsq2, mse, mse2)
            if ll is not None:
                assert abs(ll2 - ll) <= 2e-5 * abs(ll), ("table logLik", ph, h, ll, ll2)
        if res.loco[ph] is None:
            assert not os.path.exists(os.path.join(d, "out_%d.loco" % (ph + 1)))
            continue
        lines = open(os.path.join(d, "out_%d.loco" % (ph + 1))).read().splitlines()
        ids = lines[0].split()[1:]
        ref = np.array([[np.nan if t == "NA" else float(t) for t in ln.split()[1:]] for ln in lines[1:]])
        gids, got = pin.oracle_loco_rows(res, ph)
        assert ids == gids
        pin.assert_text_equal(got, ref, "pheno %d" % (ph + 1))
        ok = ~np.isnan(ref)
        worst = max(worst, float(np.max(np.abs(got[ok] - ref[ok])) / np.max(np.abs(ref[ok]))))
    extra = ""
    if skipped or len(names) != len(open(S + ".pheno").readline().split()) - 2 - (0 if not o.get("pheno_cols") else 0) and not o.get("pheno_cols"):
        # (a trait without predictions -- not converged, or dropped by --minCaseCount: regenie's --step 2 would need --phenoColList; the legs below are skipped)
        return desc + " | ok: loco max rel err %.1e (%s), regenie %.1f s, oracle %.1f s, %d trait(s) not converged in both, %d of the file's traits analysed" % (
            worst, "LOOCV" if res.use_loocv else "K-fold", t_ref, t_or, skipped, len(names)), True
    if not o["bt"] and not o.get("ct"):
        extra = ", step 2: %d statistics" % step2_qt_leg(d, S, o)
        if os.environ.get("FUZZ_BGEN"):
            extra += ", bgen: %s rows" % step2_qt_bgen_leg(d, S, g, spec, o)
    if os.environ.get("FUZZ_PGEN") and not o.get("ct"):
        extra += ", pgen: " + pgen_legs(d, S, g, spec, o, args)
    if os.environ.get("FUZZ_BGEN") == "2" and not o.get("ct"):
        extra += ", step 1 from bgen: " + step1_bgen_leg(d, S, g, spec, o, args)
    elif o.get("ct") and os.environ.get("FUZZ_BT_STEP2"):
        extra = ", step 2 --ct (score test): %d statistics" % step2_bt_leg(d, S, o)
    elif o["bt"] and os.environ.get("FUZZ_BT_STEP2"):
        extra = ", step 2 (score test): %d statistics" % step2_bt_leg(d, S, o)
        if os.environ["FUZZ_BT_STEP2"] in ("2", "3", "4"):
            extra += ", corrected: %d Firth + %d SPA rows" % step2_bt_corrections_leg(d, S, o)
        if os.environ["FUZZ_BT_STEP2"] == "3":
            extra += ", from BGEN dosages: %d Firth + %d SPA rows" % step2_bt_corrections_leg(d, S, o, bgen=(g, spec))
        if os.environ["FUZZ_BT_STEP2"] == "4":
            extra += ", from a .pgen with dosages: %d Firth + %d SPA rows" % step2_bt_corrections_leg(d, S, o, pgen=(g, spec))
    if os.environ.get("FUZZ_DRIVER") and not os.environ.get("FUZZ_DRIVER_BT_ONLY"):
        extra += " | " + driver_legs(d, args, o, len(names))
    if skipped:
        extra += ", %d trait(s) not converged in both" % skipped
    return desc + " | ok: loco max rel err %.1e (%s), regenie %.1f s, oracle %.1f s%s" % (worst, "LOOCV" if res.use_loocv else "K-fold", t_ref, t_or, extra), True


def _prep_args(S, o):
    a = []
    for k, flag in (("remove", "--remove"), ("exclude", "--exclude"), ("keep", "--keep"), ("extract", "--extract")):
        for f in o.get(k, ()):
            a += [flag, f]
    a += ["--apply-rint"] if o.get("apply_rint") else []
    a += ["--phenoColList", ",".join(o["pheno_cols"])] if o.get("pheno_cols") else []
    a += ["--covarColList", ",".join(o["covar_cols"])] if o.get("covar_cols") else []
    a += ["--cc12"] if o.get("cc12") else []
    a += ["--minCaseCount", str(o["min_case_count"])] if o.get("min_case_count") else []
    a += ["--catCovarList", ",".join(o["cat_covar"])] if o.get("cat_covar") else []
    return a


def step2_qt_leg(d, S, o):
    """regenie --step 2 --qt on the case's files with ITS OWN step-1 predictions against the oracle's score test; -> variants x traits compared"""
    from oracle import regenie_step2_qt as s2
    args = ["--step", "2", "--qt", "--bed", S, "--phenoFile", S + ".pheno", "--covarFile", S + ".covar", "--bsize", "200", "--pred", "out_pred.list"]
    args += ["--ref-first"] if o["ref_first"] else []
    args += ["--strict"] if o["strict"] else []
    args += _prep_args(S, o)
    r = subprocess.run([REGENIE] + args + ["--out", "s2"], cwd=d, capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout + r.stderr)[-600:]
    opt = orc.Step1Options(bed=S, pheno_file=S + ".pheno", covar_file=S + ".covar", bsize=200, ref_first=o["ref_first"], strict=o["strict"], test_mode=True,
                           **{k: o[k] for k in ("remove", "exclude", "keep", "extract", "pheno_cols", "covar_cols", "apply_rint", "cat_covar") if k in o})
    bim, chrom, offs, snp_ids, prep = orc.load_inputs(opt)
    bed, _ = orc.open_bed(S + ".bed", prep.n_file)
    ia = prep.ind_in_analysis
    ids = [i for i, k in zip(prep.ids, ia) if k]
    P = prep.Y.shape[1]
    loco, refs = [], []
    for ph in range(P):
        lines = open(os.path.join(d, "out_%d.loco" % (ph + 1))).read().splitlines()
        pos = {s_: k for k, s_ in enumerate(lines[0].split()[1:])}
        v = np.array([[np.nan if t == "NA" else float(t) for t in ln.split()[1:]] for ln in lines[1:]])
        loco.append(np.nan_to_num(v[:, [pos[i] for i in ids]]))
        refs.append(pin._read_regenie(os.path.join(d, "s2_%s.regenie" % prep.pheno_names[ph])))
    col = {nm: i for i, nm in enumerate(refs[0][0])}
    X, Y, mask = prep.X[ia], prep.Y[ia], prep.mask[ia].astype(np.float64)
    by_id = [{r_[col["ID"]]: r_ for r_ in refs[ph][1]} for ph in range(P)]
    ncmp = 0
    for c in sorted(set(chrom.tolist())):
        blup = np.stack([loco[ph][c - 1] for ph in range(P)], axis=1)
        res, _, scf = s2.compute_res(Y, blup * mask, mask, prep.Neff, X.shape[1], prep.scale_Y)
        sel = np.flatnonzero(chrom == c)
        G = orc.decode_bed_rows(np.asarray(bed[offs[sel]]), prep.n_file)[:, ~prep.ind_ignore][:, ia]
        if o["ref_first"]:
            G = np.where(G < 0, G, 2.0 - G)
        out = s2.score_qt_block_ref(G, X, res, mask, scf, n_samples=int((~prep.ind_ignore).sum()))
        for k in range(sel.size):
            for ph in range(P):
                r_ = by_id[ph].get(snp_ids[sel[k]])
                if r_ is None or r_[col["BETA"]] == "NA":       # filtered by regenie (low MAC) / not testable
                    continue
                beta, se, chisq, logp = (float(r_[col[nm]]) for nm in ("BETA", "SE", "CHISQ", "LOG10P"))
                assert abs(out["bhat"][k, ph] - beta) <= 5e-5 * abs(beta) + 2e-6, ("BETA", snp_ids[sel[k]], ph, out["bhat"][k, ph], beta)
                assert abs(out["se"][k, ph] - se) <= 5e-5 * abs(se), ("SE", snp_ids[sel[k]], ph, out["se"][k, ph], se)
                assert abs(out["chisq"][k, ph] - chisq) <= 1e-4 * abs(chisq) + 2e-6, ("CHISQ", snp_ids[sel[k]], ph, out["chisq"][k, ph], chisq)
                assert abs(s2.get_logp(out["chisq"][k, ph]) - logp) <= 1e-4 * abs(logp) + 2e-6, ("LOG10P", snp_ids[sel[k]], ph)
                ncmp += 1
    assert ncmp > 0
    return ncmp


def run_t2e(seed, d, S, g, spec, o):
    """--t2e: (time, event) pairs, Cox ridge at level 1 (oracle/regenie_step1_t2e.py) -- penalties, held-out deviances, selection, .loco files"""
    from oracle import regenie_step1_t2e as t2e
    from tests.util import write_t2e_pheno
    write_t2e_pheno(S + ".t2e", g, seed=spec["seed"], **spec["t2e"])
    nt = spec["t2e"]["ntraits"]
    tcols, ecols = ["T%d" % (k + 1) for k in range(nt)], ["E%d" % (k + 1) for k in range(nt)]
    args = ["--step", "1", "--bed", S, "--phenoFile", S + ".t2e", "--covarFile", S + ".covar", "--bsize", str(o["bsize"]), "--cv", str(o["cv_folds"]),
            "--l0", str(o["n_ridge_l0"]), "--l1", str(o["n_ridge_l1"]), "--t2e", "--phenoColList", ",".join(tcols), "--eventColList", ",".join(ecols)]
    args += ["--ref-first"] if o["ref_first"] else []
    pk, prep_desc = {}, ""
    if os.environ.get("FUZZ_PREP"):
        pr = draw_prep(seed, "t2e_kfold")
        for k in ("phenocol", "rint", "setl0", "setl1", "nb", "covarcol"):      # (options that do not apply to (time, event) pairs, or are drawn elsewhere)
            pr[k] = False
        pa, pk = apply_prep(S, spec, pr)
        args += pa
        prep_desc = "".join(" " + k for k in ("remove", "exclude", "keep", "extract") if pr.get(k)) + (" cat%d" % pr["levels"] if pr["cat"] else "")
    t0 = time.time()
    r = subprocess.run([REGENIE] + args + ["--out", "out"], cwd=d, capture_output=True, text=True)
    t_ref = time.time() - t0
    desc = "seed %d t2e_kfold N %d M %d chr %d traits %d bsize %d cv %d l0 %d l1 %d%s missG %.2f missing pairs %.2f decimals %d" % (
        seed, spec["N"], spec["M"], len(set(spec["chroms"])), nt, o["bsize"], o["cv_folds"], o["n_ridge_l0"], o["n_ridge_l1"], " ref-first" if o["ref_first"] else "",
        spec["miss_rate"], spec["t2e"]["missing"], spec["t2e"]["decimals"]) + prep_desc
    if r.returncode != 0:
        return desc + " | regenie itself stopped: " + (r.stdout + r.stderr).strip().splitlines()[-1][:160], None
    t0 = time.time()
    res = t2e.run_step1_t2e(orc.Step1Options(bed=S, pheno_file=S + ".t2e", covar_file=S + ".covar", bsize=o["bsize"], cv_folds=o["cv_folds"], n_ridge_l0=o["n_ridge_l0"],
                                             n_ridge_l1=o["n_ridge_l1"], ref_first=o["ref_first"], **pk), dict(zip(tcols, ecols)))
    t_or = time.time() - t0
    ref_lines = table_lines(open(os.path.join(d, "out.log")).read())
    got_lines = [ln.rstrip() for ln in res["log"]]
    assert len(ref_lines) == len(got_lines), (len(ref_lines), len(got_lines))
    for a, b in zip(ref_lines, got_lines):
        if a.startswith("phenotype"):
            assert a.split() == b.split(), (a, b)
            continue
        ma, mb = pin.T2E_RE.match(a), pin.T2E_RE.match(b)
        assert ma and mb, (a, b)
        assert abs(float(mb.group(1)) - float(ma.group(1))) <= 2e-5 * abs(float(ma.group(1))) and abs(float(mb.group(2)) - float(ma.group(2))) <= 2e-5 * abs(float(ma.group(2))), (a, b)
        assert bool(ma.group(3)) == bool(mb.group(3)), ("selected penalty", a, b)
    prep = res["prep"]
    order = [i for i in sorted(range(len(prep.ids)), key=lambda i: prep.ids[i]) if prep.ind_in_analysis[i]]
    worst = 0.0
    for tn in tcols:
        ti = prep.pheno_names.index(tn)
        ids, ref = _loco(os.path.join(d, "out_%d.loco" % (ti + 1)))
        got = res["traits"][tn]["loco"][order, :].T.copy()
        got[:, ~prep.mask[order, ti]] = np.nan
        assert ids == [prep.ids[i] for i in order]
        pin.assert_text_equal(got, ref, tn)
        ok = ~np.isnan(ref)
        worst = max(worst, float(np.max(np.abs(got[ok] - ref[ok])) / np.max(np.abs(ref[ok]))))
    extra = ""
    if os.environ.get("FUZZ_DRIVER"):
        r = subprocess.run([BIN] + args + ["--out", "drv"], cwd=d, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, "regenie-amd --step 1 --t2e: " + (r.stdout + r.stderr)[-500:]
        assert open(os.path.join(d, "drv_pred.list")).read().replace("drv_", "out_") == open(os.path.join(d, "out_pred.list")).read()
        wd, same = 0.0, 0
        for tn in tcols:
            ti = prep.pheno_names.index(tn)
            ids_r, ref = _loco(os.path.join(d, "out_%d.loco" % (ti + 1)))
            ids_g, got = _loco(os.path.join(d, "drv_%d.loco" % (ti + 1)))
            assert ids_r == ids_g and np.array_equal(np.isnan(got), np.isnan(ref))
            ok = ~np.isnan(ref)
            wd = max(wd, float(np.max(np.abs(got[ok] - ref[ok])) / np.max(np.abs(ref[ok]))))
            same += open(os.path.join(d, "out_%d.loco" % (ti + 1))).read() == open(os.path.join(d, "drv_%d.loco" % (ti + 1))).read()
        assert wd < 1e-5, "driver .loco (t2e): %.2e" % wd
        extra = " | driver: loco %.1e (%d/%d files byte-identical)" % (wd, same, nt)
    return desc + " | ok: loco max rel err %.1e, regenie %.1f s, oracle %.1f s%s" % (worst, t_ref, t_or, extra), True


def step1_bgen_leg(d, S, g, spec, o, args1):
    """--step 1 on the case's genotypes as BGEN dosages (readChunkFromBGENFileToG, Geno.cpp:1574-1699: level 0 on doubles): regenie's .loco files
    against the oracle's run with the dosages of oracle/bgen.py as input; -> files compared"""
    from oracle import bgen as obg
    from tests.util import write_synth_bgen
    if not os.path.exists(S + ".bgen"):
        write_synth_bgen(S, g, spec["chroms"], seed=spec["seed"])
    a = [x for x in args1]
    i = a.index("--bed")
    a[i:i + 2] = ["--bgen", S + ".bgen", "--sample", S + ".sample"]
    r = subprocess.run([REGENIE] + a + ["--out", "b1"], cwd=d, capture_output=True, text=True)
    if r.returncode != 0:
        return "regenie stopped: " + (r.stdout + r.stderr).strip().splitlines()[-1][:120]
    bg = obg.BgenOracle(S + ".bgen")
    rf = o["ref_first"]
    res = orc.run_step1(orc.Step1Options(bed=S, pheno_file=S + ".pheno", covar_file=S + ".covar",
                                         dosage_provider=lambda offs: np.stack([bg.dosages(int(j), rf) for j in offs]), **o))
    names = [ln.split()[0] for ln in open(os.path.join(d, "b1_pred.list"))]
    assert names == [res.prep.pheno_names[ph] for ph in range(len(res.loco)) if res.loco[ph] is not None], ("traits with predictions (bgen)", names)
    n = 0
    for ph in range(len(res.loco)):
        if res.loco[ph] is None:
            continue
        ids, ref = _loco(os.path.join(d, "b1_%d.loco" % (ph + 1)))
        gids, got = pin.oracle_loco_rows(res, ph)
        assert ids == gids
        pin.assert_text_equal(got, ref, "bgen step 1, pheno %d" % (ph + 1))
        n += 1
    return "%d files" % n + driver_same(d, a, "b1", "step1")


def pgen_legs(d, S, g, spec, o, args1):
    """The case as a PLINK2 .pgen with a 16-bit dosage track on 40 % of the calls (PgenReader::Read: the ALT dosage, --ref-first does not apply):
    --step 1 --pgen against the oracle fed by oracle/pgen.py, and for the quantitative cases --step 2 --qt --pgen with its per-trait N, A1FREQ and
    MaCH-r2 INFO columns (compute_aaf_info, Geno.cpp:3132-3141)."""
    from oracle import pgen as opg
    from oracle import regenie_step2_qt as s2
    from tests.util import write_synth_pgen
    write_synth_pgen(S + "_p", g, spec["chroms"], seed=spec["seed"], soft=0.4)
    a = [x for x in args1]
    i = a.index("--bed")
    a[i:i + 2] = ["--pgen", S + "_p"]
    r = subprocess.run([REGENIE] + a + ["--out", "p1"], cwd=d, capture_output=True, text=True)
    if r.returncode != 0:
        return "regenie stopped: " + (r.stdout + r.stderr).strip().splitlines()[-1][:120]
    po = opg.PgenOracle(S + "_p.pgen")
    res = orc.run_step1(orc.Step1Options(bed=S, pheno_file=S + ".pheno", covar_file=S + ".covar",
                                         dosage_provider=lambda offs: np.stack([po.dosages(int(j)) for j in offs]), **o))
    names = [ln.split()[0] for ln in open(os.path.join(d, "p1_pred.list"))]
    assert names == [res.prep.pheno_names[ph] for ph in range(len(res.loco)) if res.loco[ph] is not None], ("traits with predictions (pgen)", names)
    nfile = 0
    for ph in range(len(res.loco)):
        if res.loco[ph] is None:
            continue
        ids, ref = _loco(os.path.join(d, "p1_%d.loco" % (ph + 1)))
        gids, got = pin.oracle_loco_rows(res, ph)
        assert ids == gids
        pin.assert_text_equal(got, ref, "pgen step 1, pheno %d" % (ph + 1))
        nfile += 1
    out = "%d files" % nfile + driver_same(d, a, "p1", "step1")
    if o["bt"] or o.get("ct"):
        return out
    args = ["--step", "2", "--qt", "--pgen", S + "_p", "--phenoFile", S + ".pheno", "--covarFile", S + ".covar", "--bsize", "200", "--pred", "out_pred.list"]
    args += ["--ref-first"] if o["ref_first"] else []
    args += ["--strict"] if o["strict"] else []
    args += _prep_args(S, o)
    r = subprocess.run([REGENIE] + args + ["--out", "sp"], cwd=d, capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout + r.stderr)[-600:]
    opt = orc.Step1Options(bed=S, pheno_file=S + ".pheno", covar_file=S + ".covar", bsize=200, ref_first=o["ref_first"], strict=o["strict"], test_mode=True,
                           **{k: o[k] for k in ("remove", "exclude", "keep", "extract", "pheno_cols", "covar_cols", "apply_rint", "cat_covar") if k in o})
    bim, chrom, offs, snp_ids, prep = orc.load_inputs(opt)
    ia = prep.ind_in_analysis
    ids = [i for i, k in zip(prep.ids, ia) if k]
    P = prep.Y.shape[1]
    loco, rows, col = [], [], None
    for ph in range(P):
        hdr, v = _loco(os.path.join(d, "out_%d.loco" % (ph + 1)))
        pos = {s_: k for k, s_ in enumerate(hdr)}
        loco.append(np.nan_to_num(v[:, [pos[i] for i in ids]]))
        h, body = pin._read_regenie(os.path.join(d, "sp_%s.regenie" % prep.pheno_names[ph]))
        col = {nm: i for i, nm in enumerate(h)}
        rows.append({r_[col["ID"]]: r_ for r_ in body})
    X, Y, mask = prep.X[ia], prep.Y[ia], prep.mask[ia].astype(np.float64)
    keep = ~prep.ind_ignore
    ncmp = 0
    for c in sorted(set(chrom.tolist())):
        blup = np.stack([loco[ph][c - 1] for ph in range(P)], axis=1)
        res2, _, scf = s2.compute_res(Y, blup * mask, mask, prep.Neff, X.shape[1], prep.scale_Y)
        sel = np.flatnonzero(chrom == c)
        G = np.stack([po.dosages(int(offs[k]))[keep][ia] for k in sel])
        sc = s2.score_qt_block_ref(G, X, res2, mask, scf, n_samples=int(keep.sum()), zero_count_rule=True)
        for k in range(sel.size):
            obs = G[k] >= 0
            for ph in range(P):
                r_ = rows[ph].get(snp_ids[sel[k]])
                if r_ is None or r_[col["A1FREQ"]] == "NA":
                    continue
                use = obs & (mask[:, ph] > 0)
                ns, tot = int(use.sum()), float(G[k][use].sum())
                af = tot / (2 * ns)
                info = 1.0 if af in (0.0, 1.0) else (float((G[k][use] ** 2).sum()) / ns - 4 * af * af) / (2 * af * (1 - af))
                assert int(r_[col["N"]]) == ns, ("N (pgen)", snp_ids[sel[k]], ph, ns, r_[col["N"]])
                assert abs(af - float(r_[col["A1FREQ"]])) <= 1e-5 * max(af, 1e-3), ("A1FREQ (pgen)", snp_ids[sel[k]], ph, af, r_[col["A1FREQ"]])
                if r_[col["INFO"]] == "NA":
                    assert info < 0, ("INFO is NA (pgen)", snp_ids[sel[k]], ph, info)
                else:
                    assert abs(info - float(r_[col["INFO"]])) <= 2e-5 * max(abs(info), 1e-2), ("INFO (pgen)", snp_ids[sel[k]], ph, info, r_[col["INFO"]])
                if r_[col["BETA"]] == "NA":
                    continue
                beta, se, chisq = (float(r_[col[nm]]) for nm in ("BETA", "SE", "CHISQ"))
                assert abs(sc["bhat"][k, ph] - beta) <= 5e-5 * abs(beta) + 2e-6, ("BETA (pgen)", snp_ids[sel[k]], ph, sc["bhat"][k, ph], beta)
                assert abs(sc["se"][k, ph] - se) <= 5e-5 * abs(se), ("SE (pgen)", snp_ids[sel[k]], ph)
                assert abs(sc["chisq"][k, ph] - chisq) <= 1e-4 * abs(chisq) + 2e-6, ("CHISQ (pgen)", snp_ids[sel[k]], ph)
                ncmp += 1
    return out + ", %d step-2 rows" % ncmp + driver_same(d, args, "sp", "step2")


def step2_qt_bgen_leg(d, S, g, spec, o):
    """The quantitative cases on BGEN input (8-bit probabilities, 40 % of the calls smeared into genuine probabilities): regenie --step 2 --qt
    --bgen with its own predictions against the oracle on the dosages -- BETA / SE / CHISQ / LOG10P and the per-trait A1FREQ, INFO (IMPUTE) and N
    columns (parseSnpfromBGEN, Geno.cpp:2186-2330; update_trait_counts; compute_aaf_info :3110-3146)."""
    from oracle import bgen as obg
    from oracle import regenie_step2_qt as s2
    from tests.util import write_synth_bgen
    write_synth_bgen(S, g, spec["chroms"], seed=spec["seed"])
    args = ["--step", "2", "--qt", "--bgen", S + ".bgen", "--sample", S + ".sample", "--phenoFile", S + ".pheno", "--covarFile", S + ".covar", "--bsize", "200", "--pred", "out_pred.list"]
    args += ["--ref-first"] if o["ref_first"] else []
    args += ["--strict"] if o["strict"] else []
    args += _prep_args(S, o)
    filt = None
    if os.environ.get("FUZZ_BGEN_FILTER"):        # --minMAC / --minINFO: which tests regenie leaves out (compute_mac, Geno.cpp:3077-3108; compute_aaf_info :3110-3146)
        frng = np.random.default_rng(spec["seed"] + 77)
        filt = (float(frng.choice([5, 20, 60])), float(frng.choice([0.3, 0.6, 0.75])))
        args += ["--minMAC", "%g" % filt[0], "--minINFO", "%g" % filt[1]]
    r = subprocess.run([REGENIE] + args + ["--out", "sb"], cwd=d, capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout + r.stderr)[-600:]
    opt = orc.Step1Options(bed=S, pheno_file=S + ".pheno", covar_file=S + ".covar", bsize=200, ref_first=o["ref_first"], strict=o["strict"], test_mode=True,
                           **{k: o[k] for k in ("remove", "exclude", "keep", "extract", "pheno_cols", "covar_cols", "apply_rint", "cat_covar") if k in o})
    bim, chrom, offs, snp_ids, prep = orc.load_inputs(opt)
    ia = prep.ind_in_analysis
    ids = [i for i, k in zip(prep.ids, ia) if k]
    P = prep.Y.shape[1]
    loco, rows, col = [], [], None
    for ph in range(P):
        hdr, v = _loco(os.path.join(d, "out_%d.loco" % (ph + 1)))
        pos = {s_: k for k, s_ in enumerate(hdr)}
        loco.append(np.nan_to_num(v[:, [pos[i] for i in ids]]))
        h, body = pin._read_regenie(os.path.join(d, "sb_%s.regenie" % prep.pheno_names[ph]))
        col = {nm: i for i, nm in enumerate(h)}
        rows.append({r_[col["ID"]]: r_ for r_ in body})
    X, Y, mask = prep.X[ia], prep.Y[ia], prep.mask[ia].astype(np.float64)
    bg = obg.BgenOracle(S + ".bgen")
    vidx = {v["rsid"]: k for k, v in enumerate(bg.variants)}
    keep = ~prep.ind_ignore
    ncmp = 0
    for c in sorted(set(chrom.tolist())):
        blup = np.stack([loco[ph][c - 1] for ph in range(P)], axis=1)
        res, _, scf = s2.compute_res(Y, blup * mask, mask, prep.Neff, X.shape[1], prep.scale_Y)
        sel = np.flatnonzero(chrom == c)
        GI = [bg.dosages(vidx[snp_ids[k]], o["ref_first"], want_info=True) for k in sel]
        G = np.stack([gi[0][keep][ia] for gi in GI])
        E = np.stack([gi[1][keep][ia] for gi in GI])
        out = s2.score_qt_block_ref(G, X, res, mask, scf, n_samples=int(keep.sum()))
        for k in range(sel.size):
            obs = G[k] >= 0
            for ph in range(P):
                r_ = rows[ph].get(snp_ids[sel[k]])
                use = obs & (mask[:, ph] > 0)
                ns, tot = int(use.sum()), float(G[k][use].sum())
                af = tot / (2 * ns)
                info = 1.0 if af in (0.0, 1.0) else 1 - float(E[k][use].sum()) / (2 * ns * af * (1 - af))
                if filt is not None:               # the variant as a whole (all analysed samples), then the trait's own counts
                    nsa, tota = int(obs.sum()), float(G[k][obs].sum())
                    afa = tota / (2 * nsa)
                    infoa = 1.0 if afa in (0.0, 1.0) else 1 - float(E[k][obs].sum()) / (2 * nsa * afa * (1 - afa))
                    keep_row = min(tota, 2 * nsa - tota) >= filt[0] and infoa >= filt[1] and min(tot, 2 * ns - tot) >= filt[0] and info >= filt[1]
                    present = r_ is not None and r_[col["A1FREQ"]] != "NA"
                    near = min(abs(min(tota, 2 * nsa - tota) - filt[0]), abs(min(tot, 2 * ns - tot) - filt[0])) < 1e-9 or min(abs(infoa - filt[1]), abs(info - filt[1])) < 1e-9
                    assert present == keep_row or near, ("--minMAC / --minINFO", snp_ids[sel[k]], ph, present, keep_row, tota, nsa, infoa, tot, ns, info)
                if r_ is None or r_[col["A1FREQ"]] == "NA":      # (a test regenie ignores for the trait, e.g. minimum MAC: the row is all NA)
                    continue
                assert int(r_[col["N"]]) == ns, ("N", snp_ids[sel[k]], ph, ns, r_[col["N"]])
                assert abs(af - float(r_[col["A1FREQ"]])) <= 1e-5 * max(af, 1e-3), ("A1FREQ", snp_ids[sel[k]], ph, af, r_[col["A1FREQ"]])
                if r_[col["INFO"]] == "NA":                    # print_sum_stats_single (Step2_Models.cpp:2505, :2516): a negative score is printed as NA
                    assert info < 0, ("INFO is NA in regenie's file", snp_ids[sel[k]], ph, info)
                else:
                    assert info >= 0 and abs(info - float(r_[col["INFO"]])) <= 2e-5 * max(abs(info), 1e-2), ("INFO", snp_ids[sel[k]], ph, info, r_[col["INFO"]])
                if r_[col["BETA"]] == "NA":
                    continue
                beta, se, chisq, logp = (float(r_[col[nm]]) for nm in ("BETA", "SE", "CHISQ", "LOG10P"))
                assert abs(out["bhat"][k, ph] - beta) <= 5e-5 * abs(beta) + 2e-6, ("BETA", snp_ids[sel[k]], ph, out["bhat"][k, ph], beta)
                assert abs(out["se"][k, ph] - se) <= 5e-5 * abs(se), ("SE", snp_ids[sel[k]], ph, out["se"][k, ph], se)
                assert abs(out["chisq"][k, ph] - chisq) <= 1e-4 * abs(chisq) + 2e-6, ("CHISQ", snp_ids[sel[k]], ph, out["chisq"][k, ph], chisq)
                ncmp += 1
    assert ncmp > 0 or filt is not None
    return str(ncmp) + ("" if filt is None else " (--minMAC %g --minINFO %g)" % filt) + driver_same(d, args, "sb", "step2")


def step2_bt_leg(d, S, o):
    """regenie --step 2 --bt (the score test, no Firth / SPA) with ITS OWN step-1 predictions against oracle/regenie_step2_bt.py: the null logistic
    model with the LOCO offset per chromosome, compute_score_bt, get_sumstats; -> statistics compared.  Count traits (o["ct"]): --ct, the null
    Poisson model and compute_score_ct."""
    from oracle import regenie_step2_bt as bt
    from oracle import regenie_step2_qt as s2
    ct = bool(o.get("ct"))
    args = ["--step", "2", "--ct" if ct else "--bt", "--bed", S, "--phenoFile", S + ".pheno", "--covarFile", S + ".covar", "--bsize", "200", "--pred", "out_pred.list"]
    args += ["--ref-first"] if o["ref_first"] else []
    args += ["--strict"] if o["strict"] else []
    args += _prep_args(S, o)
    if os.environ.get("FUZZ_BT_MINMAC"):       # which tests are left out (compute_mac, Geno.cpp:3077-3108; the trait's own counts where its mask differs): rows of NA in both files
        args += ["--minMAC", str([10, 40, 120][sum(map(ord, os.path.basename(d))) % 3])]
    r = subprocess.run([REGENIE] + args + ["--out", "s2"], cwd=d, capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout + r.stderr)[-600:]
    opt = orc.Step1Options(bed=S, pheno_file=S + ".pheno", covar_file=S + ".covar", bsize=200, bt=not ct, ct=ct, ref_first=o["ref_first"], strict=o["strict"], test_mode=True,
                           **{k: o[k] for k in ("remove", "exclude", "keep", "extract", "pheno_cols", "covar_cols", "cc12", "min_case_count", "cat_covar") if k in o})
    bim, chrom, offs, snp_ids, prep = orc.load_inputs(opt)
    bed, _ = orc.open_bed(S + ".bed", prep.n_file)
    ia = prep.ind_in_analysis
    ids = [i for i, k in zip(prep.ids, ia) if k]
    P = prep.Y.shape[1]
    loco, rows, col = [], [], None
    for ph in range(P):
        hdr, v = _loco(os.path.join(d, "out_%d.loco" % (ph + 1)))
        pos = {s_: k for k, s_ in enumerate(hdr)}
        loco.append(np.nan_to_num(v[:, [pos[i] for i in ids]]))
        h, body = pin._read_regenie(os.path.join(d, "s2_%s.regenie" % prep.pheno_names[ph]))
        col = {nm: i for i, nm in enumerate(h)}
        rows.append({r_[col["ID"]]: r_ for r_ in body})
    X, Yraw, mask = prep.X[ia], prep.Y_raw[ia], prep.mask[ia]
    ncmp = nline = nexact = 0
    for c in sorted(set(chrom.tolist())):
        nulls = [(bt.null_poisson if ct else bt.null_logistic)(Yraw[:, ph], X, mask[:, ph], loco[ph][c - 1], opt) for ph in range(P)]
        sel = np.flatnonzero(chrom == c)
        G = orc.decode_bed_rows(np.asarray(bed[offs[sel]]), prep.n_file)[:, ~prep.ind_ignore][:, ia]
        if o["ref_first"]:
            G = np.where(G < 0, G, 2.0 - G)
        for k in range(sel.size):
            gk, flipped = bt.flip_geno(G[k])           # the minor allele is tested (BETA negated back); the sparse form's numerator is GW . yres
            g, mean, nobs = s2.mean_impute(gk)
            sparse = s2.check_sparse(g, int((~prep.ind_ignore).sum()))
            for ph in range(P):
                r_ = rows[ph].get(snp_ids[sel[k]])
                if r_ is None or r_[col["BETA"]] == "NA" or nulls[ph] is None:
                    continue
                out = (bt.score_ct if ct else bt.score_bt)(g, X, Yraw[:, ph], mask[:, ph].astype(np.float64), nulls[ph], sparse=sparse)
                if out is None:
                    continue
                out["bhat"] = -out["bhat"] if flipped else out["bhat"]
                beta, se, chisq, logp = (float(r_[col[nm]]) for nm in ("BETA", "SE", "CHISQ", "LOG10P"))
                nline += 1
                nexact += all(("%g" % float("%.6g" % v)) == ("%g" % w) for v, w in ((out["bhat"], beta), (out["se"], se), (out["chisq"], chisq)))
                assert abs(out["bhat"] - beta) <= 5e-5 * abs(beta) + 2e-6, ("BETA", snp_ids[sel[k]], ph, out["bhat"], beta)
                assert abs(out["se"] - se) <= 5e-5 * abs(se), ("SE", snp_ids[sel[k]], ph, out["se"], se)
                assert abs(out["chisq"] - chisq) <= 1e-4 * abs(chisq) + 2e-6, ("CHISQ", snp_ids[sel[k]], ph, out["chisq"], chisq)
                assert abs(s2.get_logp(out["chisq"]) - logp) <= 1e-4 * abs(logp) + 2e-6, ("LOG10P", snp_ids[sel[k]], ph)
                ncmp += 1
    assert ncmp > 0
    print("      (oracle: %d of %d rows round to regenie's printed BETA / SE / CHISQ)" % (nexact, nline), flush=True)
    if os.environ.get("FUZZ_DRIVER"):          # the product's lines beside regenie's (closed-form score test: byte-identical but for the last digit)
        r = subprocess.run([BIN] + args + ["--out", "d2"], cwd=d, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, "regenie-amd --step 2 %s: " % ("--ct" if ct else "--bt") + (r.stdout + r.stderr)[-500:]
        same = tot = 0
        for ph in range(P):
            a = open(os.path.join(d, "d2_%s.regenie" % prep.pheno_names[ph])).read().splitlines()
            b = open(os.path.join(d, "s2_%s.regenie" % prep.pheno_names[ph])).read().splitlines()
            assert a[0] == b[0] and len(a) == len(b), "driver .regenie header / line count"
            for x, y in zip(a[1:], b[1:]):
                tot += 1
                if x == y:
                    same += 1
                    continue
                tx, ty = x.split(" "), y.split(" ")
                assert len(tx) == len(ty) and tx[:5] == ty[:5], (x, y)
                for u, v in zip(tx[5:], ty[5:]):
                    assert u == v or (u != "NA" and v != "NA" and abs(float(u) - float(v)) <= 5e-5 * abs(float(v)) + 2e-6), (x, y)
        print("      (driver, score test: %d of %d lines byte-identical to regenie's)" % (same, tot), flush=True)
    return ncmp


def step2_bt_corrections_leg(d, S, o, pthresh=0.2, bgen=None, pgen=None):
    """regenie --step 2 --bt with --firth --approx and with --spa (p-value threshold 0.2: a fifth of the tests are corrected) against
    oracle/regenie_step2_bt.py: null Firth model, the 1-parameter approximate Firth fit (on the carriers only for sparse rare variants),
    the saddlepoint approximation (its fast form for sparse variants; a test regenie reports as TEST_FAIL has no root here either).
    Both programs stop their null models and these fits at a tolerance, and the drawn cases are small (a few hundred samples): the rows agree to
    ~1e-4 here, where the fixtures of tests/test_reference_pin.py (5,000+ samples) hold 3e-5; a difference of logic would be 100x that.
    -> (firth rows, spa rows)"""
    from scipy.stats import norm
    from oracle import regenie_step2_bt as bt
    from oracle import regenie_step2_qt as s2
    # bgen = (genotypes, spec): the same case as BGEN dosages (8-bit probabilities, 40 % of the calls smeared): check_sparse_G then counts the
    # non-zero DOSAGES of the coding regenie tests -- after flip_geno the entries that are not exactly 2
    src = ["--bed", S] if bgen is None else ["--bgen", S + ".bgen", "--sample", S + ".sample"]
    tag = "" if bgen is None else "b"
    if pgen is not None:
        # pgen = (genotypes, spec): the case as a .pgen with a dosage track on two fifths of the calls; check_sparse_G then takes the zeros the reader
        # counted BEFORE flip_geno (prep_snp_stats, Geno.cpp:2582-2594) while the carriers of the fast forms are those of the flipped coding
        from oracle import pgen as opg
        from tests.util import write_synth_pgen
        if not os.path.exists(S + "_p.pgen"):
            # FUZZ_PGEN_HARD: every other case is a .pgen of hard calls only (the driver then hands the library 2-bit rows, as for a .bed)
            soft = 0.0 if os.environ.get("FUZZ_PGEN_HARD") and sum(map(ord, os.path.basename(d))) % 2 else 0.4
            write_synth_pgen(S + "_p", pgen[0], pgen[1]["chroms"], seed=pgen[1]["seed"], soft=soft)
        src, tag = ["--pgen", S + "_p"], "p"
    if bgen is not None:
        from oracle import bgen as obg
        from tests.util import write_synth_bgen
        if not os.path.exists(S + ".bgen"):
            write_synth_bgen(S, bgen[0], bgen[1]["chroms"], seed=bgen[1]["seed"])
    base = ["--step", "2", "--bt"] + src + ["--phenoFile", S + ".pheno", "--covarFile", S + ".covar", "--bsize", "200", "--pred", "out_pred.list", "--pThresh", str(pthresh)]
    base += ["--ref-first"] if o["ref_first"] else []
    base += ["--strict"] if o["strict"] else []
    base += _prep_args(S, o)
    exact = os.environ.get("FUZZ_BT_EXACT") and bgen is None and pgen is None        # also --firth without --approx (a C + 1 parameter fit per flagged test)
    # FUZZ_BT_FIRTH_SE: a third of the cases print the Firth SE as |BETA| / sqrt(LRT) (--firth-se, back_correct_se: Step2_Models.cpp:2008-2009)
    firth_se = bool(os.environ.get("FUZZ_BT_FIRTH_SE")) and sum(map(ord, os.path.basename(d))) % 3 == 0
    fse, unf = (["--firth-se"] if firth_se else []), []
    # FUZZ_BT_NULLFIRTH: every other case starts its null Firth fits from the estimates Step 1 wrote (--use-null-firth, Step2_Models.cpp:899-984)
    if os.environ.get("FUZZ_BT_NULLFIRTH") and sum(map(ord, os.path.basename(d))) % 2 == 0 and os.path.exists(os.path.join(d, "out_firth.list")):
        unf = ["--use-null-firth", "out_firth.list"]
    for extra, out in ((["--firth", "--approx"] + fse + unf, "s2f" + tag), (["--spa"], "s2s" + tag)) + (((["--firth"] + fse, "s2e"),) if exact else ()):
        r = subprocess.run([REGENIE] + base + extra + ["--out", out], cwd=d, capture_output=True, text=True)
        assert r.returncode == 0, (r.stdout + r.stderr)[-600:]
    opt = orc.Step1Options(bed=S, pheno_file=S + ".pheno", covar_file=S + ".covar", bsize=200, bt=True, ref_first=o["ref_first"], strict=o["strict"], test_mode=True,
                           **{k: o[k] for k in ("remove", "exclude", "keep", "extract", "pheno_cols", "covar_cols", "cc12", "min_case_count", "cat_covar") if k in o})
    bim, chrom, offs, snp_ids, prep = orc.load_inputs(opt)
    bed, _ = orc.open_bed(S + ".bed", prep.n_file)
    ia = prep.ind_in_analysis
    ids = [i for i, k in zip(prep.ids, ia) if k]
    P = prep.Y.shape[1]
    loco, frow, srow, erow, col = [], [], [], [], None
    ne = 0
    for ph in range(P):
        if exact:
            erow.append({r_[2]: r_ for r_ in pin._read_regenie(os.path.join(d, "s2e_%s.regenie" % prep.pheno_names[ph]))[1]})
        hdr, v = _loco(os.path.join(d, "out_%d.loco" % (ph + 1)))
        pos = {s_: k for k, s_ in enumerate(hdr)}
        loco.append(v[:, [pos[i] for i in ids]])
        h, body = pin._read_regenie(os.path.join(d, "s2f%s_%s.regenie" % (tag, prep.pheno_names[ph])))
        col = {nm: i for i, nm in enumerate(h)}
        frow.append({r_[col["ID"]]: r_ for r_ in body})
        srow.append({r_[col["ID"]]: r_ for r_ in pin._read_regenie(os.path.join(d, "s2s%s_%s.regenie" % (tag, prep.pheno_names[ph])))[1]})
    X, Yraw, mask = prep.X[ia], prep.Y_raw[ia], prep.mask[ia]
    zthr = float(norm.ppf(1 - pthresh / 2))
    n_all = int((~prep.ind_ignore).sum())
    if pgen is not None:
        pgo = opg.PgenOracle(S + "_p.pgen")
    elif bgen is not None:
        bgo = obg.BgenOracle(S + ".bgen")
        vidx = {v["rsid"]: k for k, v in enumerate(bgo.variants)}
    nf = ns = nfx = 0
    for c in sorted(set(chrom.tolist())):
        nulls, offs_f, bnulls = [], [], []
        for ph in range(P):
            nl = bt.null_logistic(Yraw[:, ph], X, mask[:, ph], np.nan_to_num(loco[ph][c - 1]), opt)
            bnull = bt.firth_null(Yraw[:, ph], X, mask[:, ph], np.nan_to_num(loco[ph][c - 1]), nl["beta"]) if nl is not None else None
            nulls.append(nl)
            bnulls.append(bnull)
            offs_f.append(X @ bnull + np.nan_to_num(loco[ph][c - 1]) if bnull is not None else None)
        sel = np.flatnonzero(chrom == c)
        if bgen is None and pgen is None:
            G = orc.decode_bed_rows(np.asarray(bed[offs[sel]]), prep.n_file)[:, ~prep.ind_ignore][:, ia]
            if o["ref_first"]:
                G = np.where(G < 0, G, 2.0 - G)
        elif pgen is not None:
            G = np.stack([pgo.dosages(int(j))[~prep.ind_ignore][ia] for j in offs[sel]])
        else:
            G = np.stack([bgo.dosages(vidx[snp_ids[k]], o["ref_first"])[~prep.ind_ignore][ia] for k in sel])
        for k in range(sel.size):
            gk, flipped = bt.flip_geno(G[k])          # regenie tests the MINOR allele and negates BETA back
            sgn = -1.0 if flipped else 1.0
            g, _, _ = s2.mean_impute(gk)
            sparse = s2.check_sparse(g, n_all) if pgen is None else int(np.count_nonzero(G[k] == 0.0)) >= n_all * 0.5
            obs = gk >= 0
            for ph in range(P):
                rf, rs = frow[ph].get(snp_ids[sel[k]]), srow[ph].get(snp_ids[sel[k]])
                if rf is None or nulls[ph] is None or offs_f[ph] is None or rf[col["BETA"]] == "NA":
                    continue
                m = mask[:, ph].astype(np.float64)
                out = bt.score_bt(g, X, Yraw[:, ph], m, nulls[ph], sparse=sparse)
                if abs(out["stats"]) <= zthr * (1 + 1e-9) + 1e-9:
                    continue
                if abs(abs(out["stats"]) - zthr) < 1e-6 * zthr:          # (a tie with the threshold is decided by the last bits)
                    continue
                if rs is not None and rs[col["BETA"]] != "NA":
                    sp = bt.spa_test(out["stats"], out["denum"], out["Gres"], nulls[ph], m, carriers=np.flatnonzero(g != 0) if sparse else None)
                    if rs[-1] == "TEST_FAIL":
                        assert sp is None, ("SPA: regenie fails, the oracle finds a root", snp_ids[sel[k]], ph)
                    else:
                        assert sp is not None, ("SPA: the oracle finds no root", snp_ids[sel[k]], ph)
                        for nm, key in (("BETA", "bhat"), ("SE", "se"), ("CHISQ", "chisq"), ("LOG10P", "logp")):
                            got = sp[key] * (sgn if key == "bhat" else 1.0)
                            assert abs(got - float(rs[col[nm]])) <= 3e-4 * abs(float(rs[col[nm]])) + 1e-6, ("SPA " + nm, snp_ids[sel[k]], ph, got, rs[col[nm]], flipped)
                    ns += 1
                if exact:           # fit_firth_logistic_snp (Step2_Models.cpp:1062-1156): the design [covariates | g], the LOCO prediction as offset
                    re_ = erow[ph].get(snp_ids[sel[k]])
                    if re_ is not None and re_[col["BETA"]] != "NA" and re_[-1] != "TEST_FAIL":
                        ex = bt.exact_firth(g, X, Yraw[:, ph], m, loco[ph][c - 1], bnulls[ph])
                        assert ex is not None, ("exact Firth: no fit", snp_ids[sel[k]], ph)
                        beta, se, chisq = (float(re_[col[nm]]) for nm in ("BETA", "SE", "CHISQ"))
                        assert abs(sgn * ex["bhat"] - beta) <= 2e-3 * se * se + 5e-4 * abs(beta) + 5e-6, ("exact Firth BETA", snp_ids[sel[k]], ph, sgn * ex["bhat"], beta, se)
                        want_se = abs(ex["bhat"]) / np.sqrt(ex["chisq"]) if firth_se and ex["chisq"] > 0 else ex["se"]
                        assert abs(want_se - se) <= (4e-3 if firth_se else 1e-3) * se, ("exact Firth SE", snp_ids[sel[k]], ph, want_se, se)
                        assert abs(ex["chisq"] - chisq) <= 3e-3 * abs(chisq) + 1e-4, ("exact Firth CHISQ", snp_ids[sel[k]], ph, ex["chisq"], chisq)
                        ne += 1
                if rf[-1] == "TEST_FAIL":
                    continue
                tq = float(gk[obs & (m > 0)].sum())
                nq = int((obs & (m > 0)).sum())
                fo = bt.approx_firth(g, X, Yraw[:, ph], m, nulls[ph], offs_f[ph], sparse=sparse, mac=min(tq, 2 * nq - tq))
                assert fo is not None, ("approximate Firth: no fit", snp_ids[sel[k]], ph)
                beta, se, chisq = (float(rf[col[nm]]) for nm in ("BETA", "SE", "CHISQ"))
                assert abs(sgn * fo["bhat"] - beta) <= 2e-3 * se * se + 3e-4 * abs(beta) + 5e-6, ("Firth BETA", snp_ids[sel[k]], ph, sgn * fo["bhat"], beta, se, flipped)
                want_se = abs(fo["bhat"]) / np.sqrt(fo["chisq"]) if firth_se and fo["chisq"] > 0 else fo["se"]
                assert abs(want_se - se) <= (4e-3 if firth_se else 5e-4) * se + 1e-6, ("Firth SE", snp_ids[sel[k]], ph, want_se, se, firth_se)
                assert abs(fo["chisq"] - chisq) <= 3e-3 * abs(chisq) + 5e-5, ("Firth CHISQ", snp_ids[sel[k]], ph, fo["chisq"], chisq)
                nf += 1
                nfx += all(("%g" % float("%.6g" % v)) == ("%g" % w_) for v, w_ in ((sgn * fo["bhat"], beta), (want_se, se), (fo["chisq"], chisq)))
    if os.environ.get("FUZZ_DRIVER"):
        # the product's corrected rows beside regenie's (both stop their fits at a tolerance: the bars of the oracle comparison above).  Variants
        # whose counted allele is the major one are where the carriers of the fast forms are those of 2 - g (flip_geno).
        ndrv = nsame = 0
        for extra, out, ref in ((["--firth", "--approx"] + fse + unf, "d2f" + tag, "s2f" + tag), (["--spa"], "d2s" + tag, "s2s" + tag)) + (((["--firth"] + fse, "d2e", "s2e"),) if exact else ()):
            r = subprocess.run([BIN] + base + extra + ["--out", out], cwd=d, capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, "regenie-amd --step 2 --bt %s: " % extra[0] + (r.stdout + r.stderr)[-500:]
            for ph in range(P):
                h, a = pin._read_regenie(os.path.join(d, "%s_%s.regenie" % (out, prep.pheno_names[ph])))
                h2, b = pin._read_regenie(os.path.join(d, "%s_%s.regenie" % (ref, prep.pheno_names[ph])))
                assert h == h2 and len(a) == len(b), ("driver rows", extra, ph, len(a), len(b))
                for x, y in zip(a, b):
                    nsame += x == y
                    assert x[:5] == y[:5] and x[-1] == y[-1], ("driver row", extra, x, y)
                    for nm in ("BETA", "SE", "CHISQ", "LOG10P"):
                        u, v = x[col[nm]], y[col[nm]]
                        assert (u == "NA") == (v == "NA"), ("driver NA", extra, x, y)
                        if u != "NA":
                            se = float(y[col["SE"]])
                            bar = 2e-3 * se * se + 3e-4 * abs(float(v)) + 5e-6 if nm == "BETA" else 3e-3 * abs(float(v)) + 5e-5
                            assert abs(float(u) - float(v)) <= bar, ("driver " + nm, extra, x, y)
                    ndrv += 1
        print("      (driver rows held to regenie's: %d, %d of them identical in every printed digit)" % (ndrv, nsame), flush=True)
    if exact:
        print("      (exact Firth rows compared: %d)" % ne, flush=True)
    print("      (oracle, approximate Firth: %d of %d rows round to regenie's printed BETA / SE / CHISQ)" % (nfx, nf), flush=True)
    return nf, ns


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    logf = open(sys.argv[3], "a") if len(sys.argv) > 3 else None
    bad = 0
    with tempfile.TemporaryDirectory() as work:
        t_start = time.time()
        for seed in range(first, first + count):
            if os.environ.get("FUZZ_BUDGET_S") and time.time() - t_start > float(os.environ["FUZZ_BUDGET_S"]):
                print("(time budget reached after %d cases)" % (seed - first), flush=True)
                break
            try:
                line, ok = run_one(seed, work)
            except Exception as e:      # noqa: BLE001
                route, spec, o = draw(seed)
                line, ok = "seed %d %s MISMATCH %s | spec %s | options %s\n%s" % (seed, route, repr(e)[:300], {k: v for k, v in spec.items() if k != "chroms"}, o,
                                                                                  traceback.format_exc()[-600:]), False
            bad += ok is False
            print(line, flush=True)
            if logf:
                logf.write("* " + line.split("\n")[0] + "\n")
                logf.flush()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
