#!/usr/bin/env python
"""TEST INFRASTRUCTURE.  Generates tests/golden/ref_outputs/: outputs of regenie v4.1.2 ITSELF, i.e. of
oracle/_ref/regenie = the reference's own sources compiled where they lie under /root/reference by
oracle/Makefile (reference flags -O3 -ffast-math -fopenmp; the Boost / BGEN-container / sqlite names the image
lacks come from oracle/ref_shim/).  The build reproduces the reference-held golden file
example/test_bin_out_firth_Y1.regenie (all non-Firth rows byte-identical, see tests/test_reference_pin.py).

Every case below is a regenie command on the reference's own example data (tests/golden/example/ holds verbatim
copies) or on the deterministic synthetic data of tests/util.py (cases that need more than 5,000 samples).
Stored per case: the .loco files (gzipped text), the CV table lines of the log, _pred.list phenotype names,
and for the split-l0 case the raw level-0 predictor files (full fp64).

  python tests/golden/make_ref_outputs.py [--out DIR]   # needs oracle/_ref/regenie (make -C oracle)
"""
from __future__ import annotations

import gzip
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
EX = os.path.join(HERE, "example")
OUT = os.path.join(HERE, "ref_outputs")
REGENIE = os.path.join(ROOT, "oracle", "_ref", "regenie")

# name -> (argument list with {E} = example dir, {S} = synthetic prefix, synthetic spec or None)
CASES = {
    # BASELINE configs[0]: example.bed, 2 QT phenotypes, --bsize 100 (K-fold)
    "qt_kfold_config1": (["--step", "1", "--bed", "{E}/example", "--covarFile", "{E}/covariates.txt",
                          "--phenoFile", "{E}/phenotype.txt", "--bsize", "100", "--qt"], None),
    "qt_kfold_3chr": (["--step", "1", "--bed", "{E}/example_3chr", "--covarFile", "{E}/covariates.txt",
                       "--phenoFile", "{E}/phenotype.txt", "--bsize", "100", "--qt"], None),
    # --nb: four blocks in all, taken chromosome by chromosome (set_blocks, Data.cpp:314-329): chromosome 2 keeps three of its four blocks, chromosome 3 none
    "qt_kfold_3chr_nb": (["--step", "1", "--bed", "{E}/example_3chr", "--covarFile", "{E}/covariates.txt",
                          "--phenoFile", "{E}/phenotype.txt", "--bsize", "100", "--nb", "4", "--qt"], None),
    "qt_kfold_3chr_opts": (["--step", "1", "--bed", "{E}/example_3chr", "--covarFile", "{E}/covariates.txt",
                            "--phenoFile", "{E}/phenotype.txt", "--remove", "{E}/fid_iid_to_remove.txt",
                            "--bsize", "70", "--cv", "3", "--ref-first", "--qt", "--print-prs"], None),
    "qt_loocv_3chr": (["--step", "1", "--bed", "{E}/example_3chr", "--covarFile", "{E}/covariates.txt",
                       "--phenoFile", "{E}/phenotype.txt", "--bsize", "100", "--qt", "--loocv"], None),
    # the reference's own Step-1 test command (test/test_bash.sh:62-89, docs/docs/options.md:20-33)
    "bt_loocv_refcmd": (["--step", "1", "--bed", "{E}/example", "--exclude", "{E}/snplist_rm.txt",
                         "--covarFile", "{E}/covariates.txt", "--phenoFile", "{E}/phenotype_bin.txt",
                         "--remove", "{E}/fid_iid_to_remove.txt", "--bsize", "100", "--bt", "--lowmem",
                         "--lowmem-prefix", "tmp_rg"], None),
    "bt_loocv_wNA": (["--step", "1", "--bed", "{E}/example_3chr", "--covarFile", "{E}/covariates.txt",
                      "--phenoFile", "{E}/phenotype_bin_wNA.txt", "--bsize", "100", "--bt"], None),
    # binary traits keep K-fold CV only from 5,000 analysed samples on (Data.cpp:353): synthetic data
    # (--write-null-firth: the null approximate-Firth estimates per chromosome, out_<k>.firth + out_firth.list, Data.cpp:1873-1902)
    "bt_kfold_synth": (["--step", "1", "--bed", "{S}", "--covarFile", "{S}.covar", "--phenoFile", "{S}.pheno",
                        "--bsize", "100", "--bt", "--write-null-firth"],
                       dict(M=400, N=5200, chroms=[1] * 150 + [2] * 130 + [5] * 120, P=3, seed=11, binary=True,
                            missing_pheno=0.02)),
    "qt_kfold_synth_missing": (["--step", "1", "--bed", "{S}", "--covarFile", "{S}.covar", "--phenoFile", "{S}.pheno",
                                "--bsize", "128", "--qt"],
                               dict(M=500, N=3001, chroms=[1] * 200 + [3] * 170 + [22] * 130, P=4, seed=5, binary=False,
                                    missing_pheno=0.05, miss_rate=0.01)),
    # count phenotypes for the Step-2 count-trait test (no count data in the reference's example directory).  regenie's own --step 1 --ct
    # does not converge on them ("Penalized poisson regression did not converge", and --loocv crashes), so the LOCO files that feed
    # --step 2 --ct come from a --qt run on the same file: any LOCO prediction is a valid offset of the null Poisson model
    "ct_synth": (["--step", "1", "--bed", "{S}", "--covarFile", "{S}.covar", "--phenoFile", "{S}.pheno", "--bsize", "100", "--qt"],
                 dict(M=300, N=1500, chroms=[1] * 160 + [2] * 140, P=2, seed=23, binary=False, counts=True, missing_pheno=0.03, miss_rate=0.01)),
    # count traits through Step 1 (round 4): counts drawn from a Poisson distribution whose rate carries the polygenic signal (mean 1.5) --
    # on these regenie's K-fold Poisson ridge converges (on the floor(exp(.)) counts of ct_synth it does not; its --loocv route crashes
    # on either).  Complete rows: with partially missing rows the reference's penalty grid is meaningless
    "ct_kfold_synth": (["--step", "1", "--bed", "{S}", "--covarFile", "{S}.covar", "--phenoFile", "{S}.pheno", "--bsize", "100", "--ct"],
                       dict(M=300, N=1500, chroms=[1] * 160 + [2] * 140, P=2, seed=23, binary=False, counts="poisson", missing_pheno=0.0, miss_rate=0.01)),
    # time-to-event traits on the example genotypes: time columns out of file order (outputs _1 and _3), tied times, pairs missing for one
    # trait or for both, --remove / --cv 3 / --ref-first.  example/phenotype_t2e.txt is synthetic (no such file in the reference's example
    # directory); tests/golden/make_t2e_example_pheno.py writes it
    "t2e_kfold_3chr_opts": (["--step", "1", "--bed", "{E}/example_3chr", "--covarFile", "{E}/covariates.txt", "--phenoFile", "{E}/phenotype_t2e.txt",
                             "--remove", "{E}/fid_iid_to_remove.txt", "--bsize", "70", "--cv", "3", "--ref-first", "--t2e",
                             "--phenoColList", "Relapse_T,Surv", "--eventColList", "Relapse,Died"], None),
    # time-to-event traits (--t2e, Cox ridge at level 1): two traits with tied event times and 3 % missing (time, event) pairs; the
    # phenotype file {S}.t2e comes from tests/util.py write_t2e_pheno
    "t2e_kfold_synth": (["--step", "1", "--bed", "{S}", "--covarFile", "{S}.covar", "--phenoFile", "{S}.t2e", "--bsize", "100", "--t2e",
                         "--phenoColList", "T1,T2", "--eventColList", "E1,E2"],
                        dict(M=300, N=1200, chroms=[1] * 120 + [2] * 100 + [7] * 80, P=2, seed=31, binary=False, missing_pheno=0.0, miss_rate=0.01,
                             t2e=dict(ntraits=2, missing=0.03))),
}
# the same data with the two undocumented level-1 switches of --t2e (Regenie.cpp:366-367): --t2e-event-l0 (selects the event column's level-0
# FILE in the --lowmem / --run-l1 modes; this in-memory run comes out byte-identical to the plain one) and --t2e-l1-pi6 (penalties from the
# heritability grid).  Round 5: the driver serves both, so they are entries of CASES (tests/test_reference_gpu.py runs the DRIVER on each).
ORACLE_CASES = {}   # fixtures only the oracle is held to (none since round 5: the driver serves both switches)
for _k, _x in (("t2e_kfold_synth_event_l0", "--t2e-event-l0"), ("t2e_kfold_synth_pi6", "--t2e-l1-pi6")):
    CASES[_k] = (CASES["t2e_kfold_synth"][0] + [_x], CASES["t2e_kfold_synth"][1])


def synth(prefix, spec):
    from tests.util import synth_dosages, write_plink, write_t2e_pheno
    g = synth_dosages(spec["M"], spec["N"], miss_rate=spec.get("miss_rate", 0.0), seed=spec["seed"])
    write_plink(prefix, g, spec["chroms"], P=spec["P"], seed=spec["seed"], binary=spec["binary"],
                missing_pheno=spec["missing_pheno"], counts=spec.get("counts", False))
    if spec.get("t2e"):
        write_t2e_pheno(prefix + ".t2e", g, seed=spec["seed"], **spec["t2e"])


def table_lines(log_text):
    """The per-phenotype CV table of Data::output (Data.cpp:1042-1077)."""
    keep, on = [], False
    for ln in log_text.splitlines():
        if ln.startswith("phenotype ") and ln.rstrip().endswith(":"):
            on = True
        if on and (ln.startswith("phenotype ") or ": Rsq = " in ln or ": Deviance = " in ln):
            keep.append(ln.rstrip())
    return keep


def run_case(name, args, spec, workdir):
    d = os.path.join(workdir, name)
    os.makedirs(d)
    S = os.path.join(d, "synth")
    if spec:
        synth(S, spec)
    cmd = [REGENIE] + [a.format(E=EX, S=S) for a in args] + ["--out", "out"]
    r = subprocess.run(cmd, cwd=d, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("%s failed:\n%s\n%s" % (name, r.stdout[-3000:], r.stderr[-3000:]))
    od = os.path.join(OUT, name)
    os.makedirs(od, exist_ok=True)
    log = open(os.path.join(d, "out.log")).read()
    meta = {"args": args, "synthetic": spec, "table": table_lines(log),
            "pred_list": [ln.split()[0] for ln in open(os.path.join(d, "out_pred.list"))]}
    for fn in sorted(os.listdir(d)):
        if fn.endswith(".loco") or fn.endswith(".prs") or fn.endswith(".firth"):
            with open(os.path.join(d, fn), "rb") as fi, gzip.GzipFile(os.path.join(od, fn + ".gz"), "wb", mtime=0) as fo:
                shutil.copyfileobj(fi, fo)
    json.dump(meta, open(os.path.join(od, "meta.json"), "w"), indent=1)
    return d


def split_l0_case(workdir):
    """--split-l0 / --run-l0 with --keep-l0 (test/test_bash.sh:91-138): the job files PFX_job<k>_l0_Y<ph> are the
    reference's level-0 predictors as raw doubles (Step1_Models.cpp:728-734) -- full-precision pins of ridge_level_0."""
    name = "qt_split_l0_3chr"
    d = os.path.join(workdir, name)
    os.makedirs(d)
    base = ["--bed", EX + "/example_3chr", "--covarFile", EX + "/covariates.txt", "--phenoFile", EX + "/phenotype.txt",
            "--bsize", "100", "--qt"]

    def rg(extra):
        r = subprocess.run([REGENIE, "--step", "1"] + base + extra, cwd=d, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("%s failed:\n%s\n%s" % (name, r.stdout[-3000:], r.stderr[-3000:]))
    rg(["--split-l0", "par,2", "--out", "split"])
    for j in (1, 2):
        rg(["--run-l0", "par.master,%d" % j, "--out", "split_l0_%d" % j])
    rg(["--run-l1", "par.master", "--keep-l0", "--out", "out"])
    od = os.path.join(OUT, name)
    os.makedirs(od, exist_ok=True)
    for fn in sorted(os.listdir(d)):
        if fn.endswith(".loco") or "_l0_Y" in fn or fn.endswith(".snplist") or fn.endswith(".master"):
            with open(os.path.join(d, fn), "rb") as fi, gzip.GzipFile(os.path.join(od, fn + ".gz"), "wb", mtime=0) as fo:
                shutil.copyfileobj(fi, fo)
    log = open(os.path.join(d, "out.log")).read()
    json.dump({"args": base, "table": table_lines(log)}, open(os.path.join(od, "meta.json"), "w"), indent=1)


def step2_cases(workdir, step1_dirs):
    """Step 2 single-variant tests fed by the reference's own Step-1 output: QT on the .bed (pins the Step-2 QT oracle)
    and the documented BT Firth command on example.bgen (the reference holds its output: example/test_bin_out_firth_Y1.regenie)."""
    od = os.path.join(OUT, "step2")
    os.makedirs(od, exist_ok=True)
    d = os.path.join(workdir, "step2")
    os.makedirs(d)
    os.symlink(os.path.join(EX, "example.bgen"), os.path.join(d, "ex.bgen"))     # no .bgi next to it (sqlite3 is a stand-in)
    runs = {
        "qt_bed_3chr": (step1_dirs["qt_kfold_3chr"], ["--step", "2", "--bed", EX + "/example_3chr", "--covarFile", EX + "/covariates.txt",
                                                      "--phenoFile", EX + "/phenotype.txt", "--bsize", "200", "--qt"]),
        "qt_bed_3chr_opts": (step1_dirs["qt_kfold_3chr"], ["--step", "2", "--bed", EX + "/example_3chr", "--covarFile", EX + "/covariates.txt",
                                                           "--phenoFile", EX + "/phenotype.txt", "--remove", EX + "/fid_iid_to_remove.txt",
                                                           "--ref-first", "--minMAC", "40", "--bsize", "300", "--qt"]),
        "bt_firth_bgen": (step1_dirs["bt_loocv_refcmd"], ["--step", "2", "--bgen", "ex.bgen", "--covarFile", EX + "/covariates.txt",
                                                         "--phenoFile", EX + "/phenotype_bin.txt", "--remove", EX + "/fid_iid_to_remove.txt",
                                                         "--bsize", "200", "--bt", "--firth", "--approx", "--pThresh", "0.01"]),
        "bt_firth_exact_bgen": (step1_dirs["bt_loocv_refcmd"], ["--step", "2", "--bgen", "ex.bgen", "--covarFile", EX + "/covariates.txt",
                                                               "--phenoFile", EX + "/phenotype_bin.txt", "--remove", EX + "/fid_iid_to_remove.txt",
                                                               "--bsize", "200", "--bt", "--firth", "--pThresh", "0.01"]),
        "bt_spa_bgen": (step1_dirs["bt_loocv_refcmd"], ["--step", "2", "--bgen", "ex.bgen", "--covarFile", EX + "/covariates.txt",
                                                       "--phenoFile", EX + "/phenotype_bin.txt", "--remove", EX + "/fid_iid_to_remove.txt",
                                                       "--bsize", "200", "--bt", "--spa", "--pThresh", "0.01"]),
        "bt_score_bed": (step1_dirs["bt_loocv_refcmd"], ["--step", "2", "--bed", EX + "/example", "--covarFile", EX + "/covariates.txt",
                                                        "--phenoFile", EX + "/phenotype_bin.txt", "--remove", EX + "/fid_iid_to_remove.txt",
                                                        "--bsize", "200", "--bt"]),
    }
    # phenotypes that differ in their missing values (5 %) and genotypes with missing calls (1 %): the sparse-genotype branch of
    # compute_score_qt (Step2_Models.cpp:402-413, variants with at most half of the samples non-zero) next to the dense one
    S = os.path.join(step1_dirs["qt_kfold_synth_missing"], "synth")
    runs["qt_synth_missing"] = (step1_dirs["qt_kfold_synth_missing"], ["--step", "2", "--bed", S, "--covarFile", S + ".covar", "--phenoFile", S + ".pheno",
                                                                      "--bsize", "200", "--qt"])
    # the same data as .bgen (8-bit probabilities, 40 % of the calls smeared: INFO < 1; also with --ref-first) and as .pgen with and
    # without a dosage track (tests/util.py write_synth_bgen / write_synth_pgen)
    from tests.util import synth_dosages, write_synth_bgen, write_synth_pgen
    spec = CASES["qt_kfold_synth_missing"][1]
    g = synth_dosages(spec["M"], spec["N"], miss_rate=spec.get("miss_rate", 0.0), seed=spec["seed"])
    write_synth_bgen(S, g, spec["chroms"], seed=spec["seed"])
    write_synth_pgen(S + "_d", g, spec["chroms"], seed=spec["seed"], soft=0.4)
    write_synth_pgen(S + "_h", g, spec["chroms"], seed=spec["seed"], soft=0.0)
    common = ["--covarFile", S + ".covar", "--phenoFile", S + ".pheno", "--bsize", "200", "--qt"]
    s1m = step1_dirs["qt_kfold_synth_missing"]
    runs["qt_synth_missing_bgen"] = (s1m, ["--step", "2", "--bgen", S + ".bgen", "--sample", S + ".sample"] + common)
    runs["qt_synth_missing_bgen_rf"] = (s1m, ["--step", "2", "--bgen", S + ".bgen", "--sample", S + ".sample", "--ref-first"] + common)
    runs["qt_synth_missing_strict"] = (s1m, ["--step", "2", "--bed", S, "--strict"] + common)
    runs["qt_synth_missing_bgen_mininfo"] = (s1m, ["--step", "2", "--bgen", S + ".bgen", "--sample", S + ".sample", "--minINFO", "0.75"] + common)
    runs["qt_synth_missing_pgen"] = (s1m, ["--step", "2", "--pgen", S + "_d"] + common)
    runs["qt_synth_missing_pgenhc"] = (s1m, ["--step", "2", "--pgen", S + "_h"] + common)
    # rare, sparse variants for the carriers-only form of the approximate Firth fit (MAC < 50): a second .bed for the bt_kfold_synth samples
    from tests.util import synth_rare_dosages, write_bed_bim
    Sb = os.path.join(step1_dirs["bt_kfold_synth"], "synth")
    sb = CASES["bt_kfold_synth"][1]
    write_bed_bim(Sb + "_rare", synth_rare_dosages(300, sb["N"], seed=sb["seed"], miss_rate=0.002), [1] * 100 + [2] * 100 + [5] * 100)
    shutil.copy(Sb + ".fam", Sb + "_rare.fam")
    runs["bt_firth_rare"] = (step1_dirs["bt_kfold_synth"], ["--step", "2", "--bed", Sb + "_rare", "--covarFile", Sb + ".covar", "--phenoFile", Sb + ".pheno",
                                                          "--bsize", "100", "--bt", "--firth", "--approx", "--pThresh", "0.3"])
    runs["bt_spa_rare"] = (step1_dirs["bt_kfold_synth"], ["--step", "2", "--bed", Sb + "_rare", "--covarFile", Sb + ".covar", "--phenoFile", Sb + ".pheno",
                                                        "--bsize", "100", "--bt", "--spa", "--pThresh", "0.3"])
    # the same Firth run started from the estimates Step 1 stored (--use-null-firth), writing its own (--write-null-firth in Step 2)
    runs["bt_firth_rare_usenull"] = (step1_dirs["bt_kfold_synth"], runs["bt_firth_rare"][1] + ["--use-null-firth", os.path.join(step1_dirs["bt_kfold_synth"], "out_firth.list"),
                                                                                               "--write-null-firth"])
    Sc = os.path.join(step1_dirs["ct_synth"], "synth")
    runs["ct_synth"] = (step1_dirs["ct_synth"], ["--step", "2", "--bed", Sc, "--covarFile", Sc + ".covar", "--phenoFile", Sc + ".pheno", "--bsize", "100", "--ct"])
    for name, (s1, args) in runs.items():
        r = subprocess.run([REGENIE] + args + ["--pred", os.path.join(s1, "out_pred.list"), "--out", name], cwd=d,
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("%s failed:\n%s\n%s" % (name, r.stdout[-3000:], r.stderr[-3000:]))
        for fn in sorted(os.listdir(d)):
            if (fn.startswith(name + "_Y") and fn.endswith(".regenie")) or (fn.startswith(name + "_") and fn.endswith(".firth")):
                with open(os.path.join(d, fn), "rb") as fi, gzip.GzipFile(os.path.join(od, fn + ".gz"), "wb", mtime=0) as fo:
                    shutil.copyfileobj(fi, fo)


def main():
    global OUT
    if len(sys.argv) > 2 and sys.argv[1] == "--out":      # write somewhere else (to compare with the committed fixtures)
        OUT = os.path.abspath(sys.argv[2])
    if not os.path.exists(REGENIE):
        raise SystemExit("build the reference first: make -C oracle")
    if len(sys.argv) > 2 and sys.argv[1] == "--only":      # (re)generate the Step-1 cases named after --only, leave the rest alone
        with tempfile.TemporaryDirectory() as wd:
            for name in sys.argv[2:]:
                args, spec = {**CASES, **ORACLE_CASES}[name]
                run_case(name, args, spec, wd)
                print("ok", name)
        return
    if os.path.isdir(OUT):
        shutil.rmtree(OUT)
    os.makedirs(OUT)
    with tempfile.TemporaryDirectory() as wd:
        dirs = {}
        for name, (args, spec) in {**CASES, **ORACLE_CASES}.items():
            dirs[name] = run_case(name, args, spec, wd)
            print("ok", name)
        split_l0_case(wd)
        print("ok split-l0")
        step2_cases(wd, dirs)
        print("ok step2")


if __name__ == "__main__":
    main()
