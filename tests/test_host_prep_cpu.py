"""The host preparation of `regenie-amd` -- parse_args, read_bim_fam (with the sample / variant filters), read_pheno_cov (phenotype and covariate
files, masks, rank-inverse-normal transform, dummy variables, covariate basis, null models, residualised phenotypes; regenie_amd/host/driver_common.cpp,
driver_inputs.cpp, driver_models.cpp) -- compiled with g++ into a harness WITHOUT the device library and held to the oracle's `load_inputs`
(oracle/regenie_step1.py, pinned to regenie's own runs on these very draws: tests/golden/fuzz_log.md) on the drawn cases and option sets of
tests/golden/fuzz_oracle_vs_reference.py (FUZZ_PREP = 1 .. 4: --remove / --keep / --exclude / --extract, --apply-rint, a categorical covariate,
--phenoColList / --covarColList, --cc12, --minCaseCount, --niter, --strict).  What the device then gets -- the analysed samples, the kept variants
and their file offsets, Y, the masks, the covariate basis, the null-model offsets -- is everything this run hands to level 0."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import regenie_step1 as orc
from tests.golden import fuzz_oracle_vs_reference as fz
from tests.util import synth_dosages, write_plink

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = r'''
#include "driver.h"
extern "C" const char* rg_last_error(const rg_ctx*) { return "no device library in this harness"; }
using namespace rgdrv;
static std::string g_err;
extern "C" const char* hp_error() { return g_err.c_str(); }
extern "C" void* hp_run(int argc, char** argv) {      // the start of rgdrv::run (driver_step1.cpp), up to where the device comes in
  Run* r = new Run;
  std::cout.flush(); fflush(stdout);
  const int saved = dup(1), nul = open("/dev/null", O_WRONLY);      // the run's log lines go to <out>.log only
  dup2(nul, 1); close(nul);
  auto back = [&]() { std::cout.flush(); fflush(stdout); dup2(saved, 1); close(saved); sout.f.close(); };
  try {
    r->p = parse_args(argc, argv);
    sout.f.open(r->p.out + ".log");
    read_bim_fam(*r);
    read_pheno_cov(*r);
    back();
    return r;
  } catch (const std::exception& e) { g_err = e.what(); back(); delete r; return nullptr; }
}
extern "C" void hp_free(void* h) { delete (Run*)h; }
extern "C" void hp_dims(void* h, int64_t* out) { Run* r = (Run*)h; out[0] = r->N; out[1] = r->P; out[2] = r->C; out[3] = r->n_analyzed; out[4] = r->n_file; out[5] = (int64_t)r->snp_ids.size(); }
extern "C" const double* hp_Y(void* h) { return ((Run*)h)->Y.data(); }
extern "C" const double* hp_X(void* h) { return ((Run*)h)->X.data(); }
extern "C" const uint8_t* hp_mask(void* h) { return ((Run*)h)->mask.data(); }
extern "C" const uint8_t* hp_ain(void* h) { return ((Run*)h)->ain.data(); }
extern "C" const uint8_t* hp_ignore(void* h) { return ((Run*)h)->ind_ignore.data(); }
extern "C" const uint8_t* hp_pass(void* h) { return ((Run*)h)->pheno_pass.data(); }
extern "C" const double* hp_offset(void* h) { Run* r = (Run*)h; return r->offset.empty() ? nullptr : r->offset.data(); }
extern "C" const double* hp_yraw(void* h) { Run* r = (Run*)h; return r->Yraw.empty() ? nullptr : r->Yraw.data(); }
extern "C" const double* hp_yevent(void* h) { Run* r = (Run*)h; return r->Yevent.empty() ? nullptr : r->Yevent.data(); }
extern "C" const double* hp_scale(void* h) { return ((Run*)h)->scale_Y.data(); }
extern "C" const double* hp_neff(void* h) { return ((Run*)h)->neff.data(); }
extern "C" const int64_t* hp_offs(void* h) { return ((Run*)h)->snp_offset.data(); }
extern "C" const int* hp_chrom(void* h) { return ((Run*)h)->snp_chrom.data(); }
extern "C" const char* hp_name(void* h, int q) { return ((Run*)h)->pheno_names[q].c_str(); }
extern "C" const char* hp_id(void* h, int64_t i) { return ((Run*)h)->ids[i].c_str(); }
extern "C" const char* hp_snp(void* h, int64_t i) { return ((Run*)h)->snp_ids[i].c_str(); }
'''


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("hostprep")
    src = d / "h.cpp"
    src.write_text(HARNESS)
    so = d / "libhp.so"
    host, csrc = os.path.join(ROOT, "regenie_amd", "host"), os.path.join(ROOT, "regenie_amd", "csrc")
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-w", "-I" + host] + [os.path.join(host, f) for f in ("driver_common.cpp", "driver_inputs.cpp", "driver_models.cpp")]
                       + [os.path.join(csrc, f) for f in ("pgen_api.cpp", "bgen_api.cpp")] + [str(src), "-o", str(so), "-lz", "-lpthread"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    L = C.CDLL(str(so))
    for f in ("hp_run", "hp_Y", "hp_X", "hp_mask", "hp_ain", "hp_ignore", "hp_pass", "hp_offset", "hp_yraw", "hp_yevent", "hp_scale", "hp_neff", "hp_offs", "hp_chrom"):
        getattr(L, f).restype = C.c_void_p
    for f in ("hp_error", "hp_name", "hp_id", "hp_snp"):
        getattr(L, f).restype = C.c_char_p
    return L


def _arr(ptr, n, dt):
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(dt)), shape=(n,)).copy()


def host_prep(L, args):
    argv = (C.c_char_p * (len(args) + 1))(b"regenie-amd", *[a.encode() for a in args])
    h = L.hp_run(C.c_int(len(args) + 1), argv)
    if not h:
        return None, L.hp_error().decode()
    h = C.c_void_p(h)
    dims = (C.c_int64 * 6)()
    L.hp_dims(h, dims)
    N, P, Cc, na, nf, M = [int(x) for x in dims]
    out = dict(N=N, P=P, C=Cc, n_analyzed=na, n_file=nf, M=M)
    out["Y"] = _arr(L.hp_Y(h), N * P, C.c_double).reshape(P, N).T
    out["X"] = _arr(L.hp_X(h), N * Cc, C.c_double).reshape(Cc, N).T
    out["mask"] = _arr(L.hp_mask(h), N * P, C.c_uint8).reshape(P, N).T.astype(bool)
    out["ain"] = _arr(L.hp_ain(h), N, C.c_uint8).astype(bool)
    out["ignore"] = _arr(L.hp_ignore(h), nf, C.c_uint8).astype(bool)
    out["pass"] = _arr(L.hp_pass(h), P, C.c_uint8).astype(bool)
    out["offset"] = _arr(L.hp_offset(h), N * P, C.c_double).reshape(P, N).T if L.hp_offset(h) else None
    out["yraw"] = _arr(L.hp_yraw(h), N * P, C.c_double).reshape(P, N).T if L.hp_yraw(h) else None
    out["yevent"] = _arr(L.hp_yevent(h), N * P, C.c_double).reshape(P, N).T if L.hp_yevent(h) else None
    out["scale"] = _arr(L.hp_scale(h), P, C.c_double)
    out["neff"] = _arr(L.hp_neff(h), P, C.c_double)
    out["offs"] = _arr(L.hp_offs(h), M, C.c_int64)
    out["chrom"] = _arr(L.hp_chrom(h), M, C.c_int)
    out["names"] = [L.hp_name(h, C.c_int(q)).decode() for q in range(P)]
    out["ids"] = [L.hp_id(h, C.c_int64(i)).decode() for i in range(N)]
    out["snps"] = [L.hp_snp(h, C.c_int64(i)).decode() for i in range(M)]
    L.hp_free(h)
    return out, None


def one_case(L, seed, prep_mode, work):
    """-> a description of what the case drew; asserts the driver's prepared inputs are the oracle's"""
    os.environ["FUZZ_PREP"] = str(prep_mode)
    try:
        route, spec, o = fz.draw(seed)
        if route == "bt_kfold":                         # (5,000+ samples for the reference's K-fold rule: the preparation does not depend on it)
            spec["N"] = 900 + seed % 300
        d = os.path.join(work, "p%d_c%d" % (prep_mode, seed))
        os.makedirs(d)
        S = os.path.join(d, "synth")
        g = synth_dosages(spec["M"], spec["N"], miss_rate=spec["miss_rate"], seed=spec["seed"])
        write_plink(S, g, spec["chroms"], P=spec["P"], seed=spec["seed"], binary=spec["binary"], missing_pheno=spec["missing_pheno"], counts=spec.get("counts", False))
        if route == "t2e_kfold":
            return t2e_case(L, seed, d, S, g, spec, o)
        args = ["--step", "1", "--bed", S, "--phenoFile", S + ".pheno", "--covarFile", S + ".covar", "--bsize", str(o["bsize"]), "--cv", str(o["cv_folds"]),
                "--l0", str(o["n_ridge_l0"]), "--l1", str(o["n_ridge_l1"])]
        args += ["--bt"] if o["bt"] else []
        args += ["--ct"] if o.get("ct") else []
        args += ["--loocv"] if o["loocv"] else []
        args += ["--ref-first"] if o["ref_first"] else []
        args += ["--strict"] if o["strict"] else []
        pr = fz.draw_prep(seed, route)
        pa, pk = fz.apply_prep(S, spec, pr)
        args += pa
        o = dict(o, **pk)
    finally:
        del os.environ["FUZZ_PREP"]
    desc = "%s%s" % (route, "".join(" " + k for k in ("remove", "exclude", "rint", "keep", "extract", "phenocol", "covarcol", "cc12", "mincase", "niter", "cat") if pr.get(k)))
    opt = orc.Step1Options(bed=S, pheno_file=S + ".pheno", covar_file=S + ".covar", **o)
    try:
        bim, chrom, offs, snp_ids, prep = orc.load_inputs(opt)
    except ValueError as e:
        got, err = host_prep(L, args + ["--out", os.path.join(d, "drv")])
        assert got is None, "the oracle stops (%s), the driver goes on" % e
        assert str(e).split("'")[0].strip()[:30] in err, (str(e), err)
        return desc + " | both stop: " + err[:60]
    got, err = host_prep(L, args + ["--out", os.path.join(d, "drv")])
    assert got is not None, err
    # samples and variants
    assert got["n_file"] == prep.n_file and np.array_equal(got["ignore"], prep.ind_ignore)
    assert got["ids"] == list(prep.ids) and np.array_equal(got["ain"], prep.ind_in_analysis) and got["n_analyzed"] == prep.n_analyzed
    assert got["snps"] == list(snp_ids) and np.array_equal(got["offs"], offs) and np.array_equal(got["chrom"], chrom)
    # traits
    assert got["names"] == list(prep.pheno_names)
    assert np.array_equal(got["mask"], prep.mask) and np.array_equal(got["neff"], prep.Neff)
    passed = np.ones(got["P"], bool) if prep.pheno_pass is None else np.asarray(prep.pheno_pass, bool)
    assert np.array_equal(got["pass"], passed)
    # the covariate basis: any orthonormal basis of the same space serves (the projections are what the run uses)
    assert got["C"] == prep.X.shape[1]
    Xd, Xo = got["X"], prep.X
    assert np.abs(Xd.T @ Xd - np.eye(got["C"])).max() < 1e-9
    assert np.abs(Xd @ (Xd.T @ Xo) - Xo).max() < 1e-9 * max(1.0, np.abs(Xo).max())
    assert np.allclose(got["scale"][passed], prep.scale_Y[passed], rtol=1e-9, atol=0)
    assert np.abs(got["Y"][:, passed] - prep.Y[:, passed]).max() <= 1e-9 * np.abs(prep.Y).max()
    if prep.Y_raw is not None:
        assert np.array_equal(got["yraw"], prep.Y_raw)
    if prep.offset is not None:
        m = prep.mask[:, passed]
        assert np.abs(got["offset"][:, passed][m] - prep.offset[:, passed][m]).max() <= 1e-7 * max(1.0, np.abs(prep.offset[:, passed][m]).max())
    return desc + " | N %d -> %d analysed, M %d -> %d, P %d (%d fitted), C %d" % (prep.n_file, prep.n_analyzed, spec["M"], len(snp_ids), got["P"], int(passed.sum()), got["C"])


def t2e_case(L, seed, d, S, g, spec, o):
    """--t2e: the time columns are the traits of the run, their event columns travel beside them; covariates centred and scaled, null Cox offsets"""
    from oracle import regenie_step1_t2e as t2e
    from tests.util import write_t2e_pheno
    write_t2e_pheno(S + ".t2e", g, seed=spec["seed"], **spec["t2e"])
    nt = spec["t2e"]["ntraits"]
    tcols, ecols = ["T%d" % (k + 1) for k in range(nt)], ["E%d" % (k + 1) for k in range(nt)]
    args = ["--step", "1", "--bed", S, "--phenoFile", S + ".t2e", "--covarFile", S + ".covar", "--bsize", str(o["bsize"]), "--cv", str(o["cv_folds"]),
            "--l0", str(o["n_ridge_l0"]), "--l1", str(o["n_ridge_l1"]), "--t2e", "--phenoColList", ",".join(tcols), "--eventColList", ",".join(ecols)]
    args += ["--ref-first"] if o["ref_first"] else []
    pr = fz.draw_prep(seed, "t2e_kfold")
    for k in ("phenocol", "rint", "setl0", "setl1", "nb", "covarcol"):      # (as the fuzz script's run_t2e: options that do not apply to (time, event) pairs)
        pr[k] = False
    pa, pk = fz.apply_prep(S, spec, pr)
    args += pa
    desc = "t2e_kfold%s" % "".join(" " + k for k in ("remove", "exclude", "keep", "extract", "cat") if pr.get(k))
    opt = orc.Step1Options(bed=S, pheno_file=S + ".t2e", covar_file=S + ".covar", bsize=o["bsize"], cv_folds=o["cv_folds"], n_ridge_l0=o["n_ridge_l0"],
                           n_ridge_l1=o["n_ridge_l1"], ref_first=o["ref_first"], **pk)
    tmap = dict(zip(tcols, ecols))
    prep = t2e.read_t2e(opt, tmap, orc.read_fam(S + ".fam"))
    t2e.prep_run_t2e(prep, tmap, opt)
    got, err = host_prep(L, args + ["--out", os.path.join(d, "drv")])
    assert got is not None, err
    assert got["ids"] == list(prep.ids) and np.array_equal(got["ain"], prep.ind_in_analysis) and np.array_equal(got["ignore"], prep.ind_ignore)
    assert got["names"] == sorted(tcols, key=lambda n: prep.pheno_names.index(n))
    assert got["C"] == prep.X.shape[1]
    Xd, Xo = got["X"], prep.X
    assert np.abs(Xd.T @ Xd - np.eye(got["C"])).max() < 1e-9 and np.abs(Xd @ (Xd.T @ Xo) - Xo).max() < 1e-9 * max(1.0, np.abs(Xo).max())
    for q, tn in enumerate(got["names"]):
        ti, ei = prep.pheno_names.index(tn), prep.pheno_names.index(tmap[tn])
        m = prep.mask[:, ti]
        assert np.array_equal(got["mask"][:, q], m) and got["neff"][q] == prep.Neff[ti]
        assert np.array_equal(got["yraw"][m, q], prep.Y_raw[m, ti]) and np.array_equal(got["yevent"][m, q], prep.Y_raw[m, ei])
        assert np.isclose(got["scale"][q], prep.scale_Y[ti], rtol=1e-9)
        assert np.abs(got["Y"][:, q] - prep.Y[:, ti]).max() <= 1e-9 * np.abs(prep.Y[:, ti]).max()
        assert np.abs(got["offset"][m, q] - prep.offset[m, ti]).max() <= 1e-6 * max(1.0, np.abs(prep.offset[m, ti]).max())
    return desc + " | N %d -> %d analysed, %d (time, event) pairs, C %d" % (prep.n_file, prep.n_analyzed, nt, got["C"])


NCASES = int(os.environ.get("HOST_PREP_CASES", "12"))        # per option set; HOST_PREP_CASES=120 for a long run
ROUTES = "qt_kfold,bt_loocv,ct_kfold,qt_loocv,bt_kfold,t2e_kfold"


@pytest.mark.parametrize("prep_mode", [1, 2, 4])
def test_driver_host_preparation_follows_the_oracle_on_drawn_cases(lib, tmp_path, prep_mode):
    os.environ["FUZZ_ROUTES"] = ROUTES
    try:
        seen = [one_case(lib, seed, prep_mode, str(tmp_path)) for seed in range(1, 1 + NCASES)]
    finally:
        del os.environ["FUZZ_ROUTES"]
    assert len(seen) == NCASES
    print("\n".join(seen))
