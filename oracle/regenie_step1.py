"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

A numpy/fp64 restatement of regenie's `--step 1` path (whole-genome stacked ridge
regression -> LOCO predictors).  Every function cites the reference lines it
follows (paths relative to /root/reference, regenie v4.1.2).  Linear algebra goes
through numpy -> OpenBLAS/LAPACK, i.e. the same class of backend the upstream
release binaries route Eigen to (Makefile:87-113).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module, and there only as the checker / the timed CPU baseline.

Parity pin: PINNED against regenie itself.  oracle/Makefile compiles the
reference's own sources into oracle/_ref/regenie (which reproduces the reference-held
example/test_bin_out_firth_Y1.regenie); tests/test_reference_pin.py requires this file to
reproduce that binary's outputs (fixtures: tests/golden/ref_outputs/, generator
tests/golden/make_ref_outputs.py) on every Step-1 route -- QT K-fold, QT LOOCV, BT LOOCV,
BT K-fold, missing data, sample / variant filters -- to the 6 printed digits of the .loco
files and CV tables, and the level-0 predictors to 1e-11 through the --run-l0 job files.
`tests/test_oracle_golden.py` keeps the reference's own test answers (`0.4504 ... min value`,
test/test_bash.sh:87; split-l0 == single run, test/test_bash.sh:91-138).
"""
from __future__ import annotations

import gzip
import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

MISSING = -999.0            # Regenie.hpp:215 missing_value_double
ETAMINTHR, ETAMAXTHR = -30.0, 30.0   # Step1_Models.hpp:30-31
NUMTOL = 1e-6               # Regenie.hpp numtol
NUMTOL_EPS = 10 * np.finfo(np.float64).eps   # Regenie.hpp:225
L1_RIDGE_TOL = 1e-4         # Regenie.hpp:289
L1_RIDGE_EPS = 1e-5         # Regenie.hpp:290
EIGEN_VAL_REL_TOL = 1e-15   # Regenie.hpp:227


# --------------------------------------------------------------------------
# options (the Step-1 subset of struct param, Regenie.hpp:181-437)
# --------------------------------------------------------------------------
@dataclass
class Step1Options:
    bed: str = ""
    pheno_file: str = ""
    covar_file: str = ""
    out: str = "regenie_out"
    bsize: int = 1000
    n_block: int = 0                 # --nb: total number of blocks (0 = as many as the block size gives)
    bt: bool = False                 # --bt (trait_mode 1); default --qt
    ct: bool = False                 # --ct (trait_mode 2): count phenotypes, Poisson level 1
    cv_folds: int = 5                # --cv
    loocv: bool = False              # --loocv
    n_ridge_l0: int = 5              # --l0
    n_ridge_l1: int = 5              # --l1
    setl0: Optional[Sequence[float]] = None
    setl1: Optional[Sequence[float]] = None
    keep: Sequence[str] = ()
    remove: Sequence[str] = ()
    extract: Sequence[str] = ()
    exclude: Sequence[str] = ()
    pheno_cols: Sequence[str] = ()   # --phenoCol / --phenoColList
    covar_cols: Sequence[str] = ()
    cat_covar: Sequence[str] = ()    # --catCovarList: categorical covariates -> K-1 dummy columns
    max_cat_levels: int = 10         # --maxCatLevels
    apply_rint: bool = False         # --apply-rint (ignored with --bt, Regenie.cpp:432)
    strict: bool = False
    test_mode: bool = False          # --step 2: rm_missing_qt (Regenie.cpp:1086) keeps a QT's missing values masked (Pheno.cpp:328)
    ref_first: bool = False
    nchrom: int = 23                 # --nauto + 1
    min_case_count: int = 10
    cc12: bool = False
    niter_max: int = 50              # logistic null (Regenie.hpp niter_max)
    t2e_event_l0: bool = False      # --t2e-event-l0: which level-0 FILE --lowmem / --run-l1 read for a time-to-event trait (Step1_Models.cpp:2259-2261); no effect in memory
    t2e_l1_pi6: bool = False        # --t2e-l1-pi6: level-1 penalties from the heritability grid, L (1 - h) / h x 6 / pi^2 (Step1_Models.cpp:2106-2110)
    niter_max_ridge: int = 100
    niter_max_line_search: int = 25
    niter_max_line_search_ridge: int = 100
    chunk_mb: int = 1000
    use_rel_path: bool = False
    print_prs: bool = False
    force_step1: bool = False
    # --split-l0 / --run-l0 analogue: global M used for lambda (Data.cpp:607)
    parallel_nGeno: Optional[int] = None
    # dosage input (pgen dosage_mode): callable(file_offsets) -> (len, N_file) float64, ALT dosages in [0, 2], -3 = missing
    # (what PgenReader::Read returns); the .bed of `bed` is then only used for its .bim / .fam
    dosage_provider: Optional[object] = None


def set_ridge_params(n: int) -> np.ndarray:
    """Regenie.cpp:1497-1508 -- {0.01, 1/(n-1), ..., 0.99}."""
    if n < 2:
        raise ValueError("number of ridge parameters must be at least 2 (=%d)" % n)
    v = np.arange(n, dtype=np.float64) * (1.0 / (n - 1))
    v[0], v[-1] = 0.01, 0.99
    return v


# --------------------------------------------------------------------------
# text / PLINK readers (Geno.cpp:518-610 read_bim, :643-690 read_fam, :735-751 prep_bed)
# --------------------------------------------------------------------------
def chr_str_to_int(s: str, nchrom: int) -> int:
    """Regenie.cpp:1583-1594."""
    if s.startswith("chr"):
        s = s[3:]
    if s[:1].isdigit():
        digits = ""
        for ch in s:
            if ch.isdigit():
                digits += ch
            else:
                break
        c = int(digits)
        if 1 <= c <= nchrom:
            return c
    elif s in ("X", "XY", "Y", "PAR1", "PAR2"):
        return nchrom
    return -1


def _open_text(path: str):
    """Files::openForRead (Files.cpp:38-64, :139-160): gzip only when the name ends in .gz AND the magic bytes match."""
    if path.endswith(".gz"):
        with open(path, "rb") as fh:
            if fh.read(2) == b"\x1f\x8b":
                return gzip.open(path, "rt")
    return open(path)


@dataclass
class Bim:
    chrom: np.ndarray       # int per variant kept
    ids: List[str]
    offset: np.ndarray      # row index in the bed file
    chr_read: List[int]     # chromosomes in file order


def read_bim(path: str, nchrom: int = 23) -> Bim:
    chrom, ids, offs, chr_read = [], [], [], []
    minchr = 0
    with open(path) as fh:
        for lineno, line in enumerate(fh):
            t = line.split()
            if len(t) < 6:
                raise ValueError("incorrectly formatted bim file at line %d" % (lineno + 1))
            c = chr_str_to_int(t[0], nchrom)
            if c == -1:
                raise ValueError("unknown chromosome code in bim file at line %d" % (lineno + 1))
            if not chr_read or c != chr_read[-1]:
                chr_read.append(c)
                if c <= minchr:
                    raise ValueError("chromosomes in bim file are not in ascending order.")
                minchr = c
            chrom.append(c)
            ids.append(t[1])
            offs.append(lineno)
    return Bim(np.asarray(chrom, np.int64), ids, np.asarray(offs, np.int64), chr_read)


def read_fam(path: str) -> List[str]:
    """Returns the FID_IID keys in file order (Geno.cpp:662)."""
    ids = []
    seen = set()
    with open(path) as fh:
        for lineno, line in enumerate(fh):
            t = line.split()
            if len(t) < 6:
                raise ValueError("incorrectly formatted fam file at line %d" % (lineno + 1))
            k = t[0] + "_" + t[1]
            if k in seen:
                raise ValueError("duplicate individual in fam file : FID_IID=" + k)
            if t[4] not in ("0", "1", "2"):
                raise ValueError("unrecognized sex code in file : '%s'" % t[4])
            seen.add(k)
            ids.append(k)
    return ids


def read_id_files(paths: Sequence[str]) -> set:
    """--keep/--remove files: >=2 whitespace columns FID IID, no header (Geno.cpp:1382-1441)."""
    out = set()
    for p in paths:
        with _open_text(p) as fh:
            for line in fh:
                t = line.split()
                if len(t) < 2:
                    raise ValueError("incorrectly formatted file: " + p)
                out.add(t[0] + "_" + t[1])
    return out


def read_snp_files(paths: Sequence[str]) -> set:
    out = set()
    for p in paths:
        with _open_text(p) as fh:
            for line in fh:
                t = line.split()
                if t:
                    out.add(t[0])
    return out


# bed 2-bit code -> value; Geno.cpp:2833-2856 buildLookupTable: maptogeno = {2,-3,1,0}
_BED_MAP = np.array([2.0, -3.0, 1.0, 0.0])
_BED_LUT = np.empty((256, 4), np.float64)
for _b in range(256):
    for _j in range(4):
        _BED_LUT[_b, _j] = _BED_MAP[(_b >> (2 * _j)) & 3]


def open_bed(path: str, n_file: int) -> Tuple[np.memmap, int]:
    """Geno.cpp:735-751 prep_bed: magic 6c 1b 01, SNP-major, ceil(N/4) bytes per SNP."""
    with open(path, "rb") as fh:
        magic = fh.read(3)
    if magic != b"\x6c\x1b\x01":
        raise ValueError("invalid bed file format.")
    bpr = (n_file + 3) // 4
    mm = np.memmap(path, dtype=np.uint8, mode="r", offset=3)
    if mm.size % bpr != 0:
        raise ValueError("bed file size does not match fam/bim")
    return mm.reshape(-1, bpr), bpr


def decode_bed_rows(rows: np.ndarray, n_file: int) -> np.ndarray:
    """packed (bs, ceil(N/4)) uint8 -> (bs, N_file) float64 with -3 for missing."""
    g = _BED_LUT[rows]                        # bs x bpr x 4
    return g.reshape(rows.shape[0], -1)[:, :n_file]


def read_chunk_from_bed(rows: np.ndarray, n_file: int, ind_ignore: np.ndarray,
                        ind_in_analysis: np.ndarray, ref_first: bool = False) -> np.ndarray:
    """Geno.cpp:1702-1769 readChunkFromBedFileToG: decode, drop ignored samples,
    per-SNP mean over analysed non-missing, missing->mean, non-analysed->0.
    Returns Gmat (bs x N) float64."""
    g = decode_bed_rows(rows, n_file)
    if ind_ignore is not None and ind_ignore.any():
        g = g[:, ~ind_ignore]
    miss = g == -3
    if ref_first:
        g = np.where(miss, g, 2 - g)
    ok = (~miss) & ind_in_analysis[None, :]
    total = np.where(ok, g, 0.0).sum(axis=1)
    ns = ok.sum(axis=1)
    with np.errstate(invalid="ignore", divide="ignore"):
        mu = total / ns
    g = np.where(miss, mu[:, None], g)         # mean_impute_g (Geno.cpp:3183-3188)
    g = np.where(ind_in_analysis[None, :], g, 0.0)
    return g


def read_chunk_from_dosages(g: np.ndarray, ind_ignore: np.ndarray, ind_in_analysis: np.ndarray) -> np.ndarray:
    """Geno.cpp:1773-1822 readChunkFromPGENFileToG in dosage_mode: Read() rows (the reader already dropped the ignored
    samples), a value outside [-3, 2] is an error, total = mean over analysed non-missing, mean_impute_g."""
    g = np.asarray(g, dtype=np.float64)
    if ind_ignore is not None and ind_ignore.any():
        g = g[:, ~ind_ignore]
    if ((g < -3) | (g > 2)).any():
        raise ValueError("there is a variant in the block that has a value not in [0,2] or missing")
    miss = g == -3
    ok = (~miss) & ind_in_analysis[None, :]
    with np.errstate(invalid="ignore", divide="ignore"):
        mu = np.where(ok, g, 0.0).sum(axis=1) / ok.sum(axis=1)
    g = np.where(miss, mu[:, None], g)
    return np.where(ind_in_analysis[None, :], g, 0.0)


# --------------------------------------------------------------------------
# phenotype / covariate prep (Pheno.cpp)
# --------------------------------------------------------------------------
def convert_double(tok: str) -> float:
    """Regenie.cpp:1663-1675."""
    if tok == "NA" or tok in ("nan", "inf"):
        return MISSING
    return float(tok)


@dataclass
class Prepared:
    """Everything the level-0/level-1 math needs (phenodt + filter + param subset)."""
    ids: List[str]                     # FID_IID of the N kept samples, file order
    n_file: int
    ind_ignore: np.ndarray             # (N_file,) bool
    ind_in_analysis: np.ndarray        # (N,) bool
    pheno_names: List[str]
    Y: np.ndarray                      # (N,P) residualised + scaled (phenotypes)
    Y_raw: Optional[np.ndarray]        # (N,P) raw 0/1 for BT (phenotypes_raw)
    mask: np.ndarray                   # (N,P) bool masked_indivs
    X: np.ndarray                      # (N,C) orthonormal basis new_cov
    Neff: np.ndarray                   # (P,)
    scale_Y: np.ndarray                # (P,)
    ncov: int
    n_analyzed: int
    offset: Optional[np.ndarray] = None        # (N,P) BT null-logistic offsets
    pheno_pass: Optional[np.ndarray] = None


def read_pheno_and_cov(opt: Step1Options, fam_ids: List[str]) -> Prepared:
    """Pheno.cpp:50-146 read_pheno_and_cov (+ pheno_read :148-364, covariate_read :573-808,
    setMasks :810-841, pheno_impute_miss :1903-1935) and Geno.cpp:1263-1341."""
    n_file = len(fam_ids)
    # ---- check_samples_include_exclude (Geno.cpp:1263-1341)
    if opt.remove:
        rm = read_id_files(opt.remove)
        keep_mask = np.array([k not in rm for k in fam_ids])
    elif opt.keep:
        kp = read_id_files(opt.keep)
        keep_mask = np.array([k in kp for k in fam_ids])
    else:
        keep_mask = np.ones(n_file, bool)
    if not keep_mask.any():
        raise ValueError("no samples remaining in the analysis.")
    ind_ignore = ~keep_mask
    ids = [k for k, m in zip(fam_ids, keep_mask) if m]
    idx = {k: i for i, k in enumerate(ids)}
    N = len(ids)
    trait_mode = 1 if opt.bt else (2 if opt.ct else 0)

    # ---- pheno_read
    with _open_text(opt.pheno_file) as fh:
        lines = fh.read().splitlines()
    hdr = lines[0].rstrip("\r").split()
    if len(hdr) < 2:
        raise ValueError("header of phenotype file has too few columns.")
    if hdr[0] != "FID" or hdr[1] != "IID":
        raise ValueError("header of phenotype file must start with: FID IID.")
    keep_cols = [True] * (len(hdr) - 2)
    if opt.pheno_cols:
        keep_cols = [h in set(opt.pheno_cols) for h in hdr[2:]]
    names = [h for h, k in zip(hdr[2:], keep_cols) if k]
    P = len(names)
    if P < 1:
        raise ValueError("need at least one phenotype.")
    strict = opt.strict or (P == 1)            # Pheno.cpp:198
    Y = np.zeros((N, P))
    mask = np.ones((N, P), bool)
    Yraw = np.zeros((N, P)) if trait_mode else None
    in_pheno = np.zeros(N, bool)
    for line in lines[1:]:
        t = line.split()
        if not t:
            continue
        if len(t) != 2 + len(keep_cols):
            raise ValueError("incorrectly formatted phenotype file.")
        k = t[0] + "_" + t[1]
        if k not in idx:
            continue
        i = idx[k]
        if in_pheno[i]:
            raise ValueError("individual appears more than once in phenotype file: FID=%s IID=%s" % (t[0], t[1]))
        in_pheno[i] = True
        all_miss = True
        ip = 0
        for j, kc in enumerate(keep_cols):
            if not kc:
                continue
            v = convert_double(t[2 + j])
            if trait_mode == 1:
                if opt.cc12 and v != MISSING:
                    v -= 1
                Yraw[i, ip] = v
                if v != 0 and v != 1:
                    if v != MISSING:
                        raise ValueError("a phenotype value is not 0/1/NA for individual: FID=%s IID=%s Y=%s" % (t[0], t[1], t[2 + j]))
                    mask[i, ip] = False
            elif trait_mode == 2:                      # Pheno.cpp:298, :313-320: counts must be non-negative
                Yraw[i, ip] = v
                if v < 0:
                    if v != MISSING:
                        raise ValueError("a phenotype value is <0 for individual: FID=%s IID=%s Y=%s" % (t[0], t[1], t[2 + j]))
                    mask[i, ip] = False
            Y[i, ip] = v
            if v != MISSING:
                all_miss = False
            else:
                if opt.test_mode and trait_mode == 0:   # Pheno.cpp:328: test_mode && rm_missing_qt
                    mask[i, ip] = False
                if strict:
                    mask[i, :] = False
                    all_miss = True
                    break
            ip += 1
        if all_miss:
            in_pheno[i] = False
    mask &= in_pheno[:, None]
    nobs = mask.sum(axis=0)
    if (nobs == 0).any():
        raise ValueError("all individuals have missing/invalid values for phenotype '%s'." % names[int(np.argmin(nobs))])
    # rm_phenoCols (Pheno.cpp:528-570): drop BT columns with too few cases
    if trait_mode == 1:
        ncases = (Yraw == 1).sum(axis=0)      # `(phenotypes_raw == 1).colwise().count()`: NOT masked -- a row --strict dropped still counts for the traits read before its first missing value
        keepp = ncases >= opt.min_case_count
        if not keepp.all():
            Y, Yraw, mask = Y[:, keepp], Yraw[:, keepp], mask[:, keepp]
            names = [n for n, k in zip(names, keepp) if k]
            P = len(names)
            if P < 1:
                raise ValueError("all phenotypes have less than %d cases." % opt.min_case_count)
            in_pheno &= mask.any(axis=1) if not strict else in_pheno

    # ---- covariates (intercept + quantitative columns; Pheno.cpp:573-808)
    X = np.ones((N, 1))
    in_cov = np.ones(N, bool) if not opt.covar_file else np.zeros(N, bool)
    if opt.covar_file:
        with _open_text(opt.covar_file) as fh:
            lines = fh.read().splitlines()
        hdr = lines[0].rstrip("\r").split()
        if hdr[0] != "FID" or hdr[1] != "IID":
            raise ValueError("header of covariate file must start with: FID IID.")
        # cov_colKeep_names (Regenie.cpp:591-619): name -> quantitative?; --catCovarList names are kept as well
        colmap = {h: True for h in opt.covar_cols}
        colmap.update({h: False for h in opt.cat_covar})
        kc = []
        for h in hdr[2:]:                                   # Pheno.cpp:599-617
            if not opt.covar_cols and h not in colmap:
                colmap[h] = True
                keep = True
            else:
                keep = h in colmap
            if keep and h in names:                         # a covariate that is an analysed phenotype is ignored
                keep = False
                del colmap[h]
            kc.append(keep)
        nc = sum(kc)
        if len(colmap) != nc:
            raise ValueError("not all covariates specified are found in the covariate file.")
        is_cat = [not colmap[h] for h, k in zip(hdr[2:], kc) if k]
        cat_names = [h for h, k in zip(hdr[2:], kc) if k]
        levels = [dict() for _ in is_cat]                   # convertNumLevel (Regenie.cpp:1720-1735): order of appearance
        X = np.zeros((N, 1 + nc))
        X[:, 0] = 1.0
        for line in lines[1:]:
            t = line.split()
            if not t:
                continue
            k = t[0] + "_" + t[1]
            if k not in idx:
                continue
            i = idx[k]
            if in_cov[i]:
                raise ValueError("individual appears more than once in covariate file: FID=%s IID=%s" % (t[0], t[1]))
            in_cov[i] = True
            ic = 0
            for j, kk in enumerate(kc):
                if not kk:
                    continue
                if is_cat[ic]:
                    tok = t[2 + j]
                    if tok in ("NA", "nan", "inf"):
                        v = MISSING
                    else:
                        v = float(levels[ic].setdefault(tok, len(levels[ic])))
                else:
                    v = convert_double(t[2 + j])
                X[i, 1 + ic] = v
                if v == MISSING:
                    in_cov[i] = False
                    break
                ic += 1
        if not in_cov.any():
            raise ValueError("none of the individuals have covariate data (check sample IDs across files)")
        X *= in_cov[:, None]
        if any(is_cat):                                     # dummies (Pheno.cpp:716-783, check_categories :985-1011, get_dummies)
            cols = [X[:, 0]]
            for ic in range(nc):
                col = X[:, 1 + ic]
                if not is_cat[ic]:
                    cols.append(col)
                    continue
                if len(levels[ic]) > opt.max_cat_levels:
                    raise ValueError("too many categories for covariate: %s (=%d). Either use '--maxCatLevels' or combine categories."
                                     % (cat_names[ic], len(levels[ic])))
                for lvl in range(1, int(col.max()) + 1):    # level 0 goes to the intercept
                    cols.append((col == lvl).astype(np.float64))
            X = np.stack(cols, axis=1)

    # ---- Pheno.cpp:101 + setMasks :810-841
    ain = in_pheno & in_cov
    ain &= mask.all(axis=1) if strict else mask.any(axis=1)
    mask &= ain[:, None]
    Y = Y * ain[:, None]
    if Yraw is not None:
        Yraw = Yraw * ain[:, None]
    X = X * ain[:, None]
    n_analyzed = int(ain.sum())
    if n_analyzed < 1:
        raise ValueError("sample size cannot be < 1.")
    Neff = mask.sum(axis=0).astype(np.float64)
    if X.shape[1] >= N:
        raise ValueError("Number of covariates is greater than sample size!")

    # ---- apply_rint / rint_pheno (Pheno.cpp:111-115, :1937-2010): ranks with ties averaged -> normal quantiles
    if opt.apply_rint and not opt.bt:
        from scipy.special import ndtri
        for j in range(P):
            sel = np.flatnonzero((Y[:, j] != MISSING) & mask[:, j])
            vals = Y[sel, j]
            order = np.argsort(vals, kind="stable")
            ranks = np.empty(sel.size)
            sv = vals[order]
            a = 0
            while a < sel.size:
                b = a + 1
                while b < sel.size and sv[b] == sv[a]:
                    b += 1
                ranks[order[a:b]] = (a + 1) + (b - a - 1) / 2.0
                a = b
            Y[sel, j] = ndtri((ranks - 3 / 8.0) / (sel.size - 2 * (3 / 8.0) + 1))

    # ---- pheno_impute_miss (Pheno.cpp:1903-1935)
    for j in range(P):
        if trait_mode == 0:
            nm = Y[:, j] != MISSING
            total = Y[nm, j].sum()
            ns = (ain & nm).sum()
            Y[:, j] = np.where(nm, Y[:, j], total / ns)
        else:
            total = Y[mask[:, j], j].sum() / mask[:, j].sum()
            Y[:, j] = np.where(mask[:, j], Y[:, j], total)
    Y = Y * mask

    return Prepared(ids=ids, n_file=n_file, ind_ignore=ind_ignore, ind_in_analysis=ain,
                    pheno_names=names, Y=Y, Y_raw=Yraw, mask=mask, X=X, Neff=Neff,
                    scale_Y=np.ones(P), ncov=X.shape[1], n_analyzed=n_analyzed,
                    pheno_pass=np.ones(P, bool))


def get_basis(X: np.ndarray) -> Tuple[np.ndarray, int]:
    """Pheno.cpp:1660-1681 getBasis: X <- X V D^-1/2 keeping eigvals > 1e-15 * max."""
    xtx = X.T @ X
    D, V = np.linalg.eigh(xtx)
    nz = int((D > D[-1] * EIGEN_VAL_REL_TOL).sum())
    Xn = X @ V[:, -nz:]
    Xn = Xn / np.sqrt(D[-nz:])[None, :]
    return Xn, nz


def get_pvec(eta: np.ndarray, eps: float = NUMTOL_EPS) -> np.ndarray:
    """Step1_Models.cpp:1799-1806."""
    with np.errstate(over="ignore"):
        p = 1 - 1 / (np.exp(eta) + 1)
    p = np.where(eta < ETAMINTHR, eps / (1 + eps), p)
    p = np.where(eta > ETAMAXTHR, 1 / (1 + eps), p)
    return p


def get_wvec(p: np.ndarray, mask: np.ndarray) -> Tuple[np.ndarray, bool]:
    """Step1_Models.cpp:1760-1783."""
    w = np.where(mask, p * (1 - p), 1.0)
    return w, bool((w == 0).any())


def logist_dev(y: np.ndarray, p: np.ndarray, mask: np.ndarray) -> float:
    """Step1_Models.cpp:1820-1828 (+ compute_log_lik_bern :1841-1844)."""
    with np.errstate(divide="ignore"):
        ll = -np.where(y == 0, np.log(1 - p), np.log(p))
    return 2.0 * float(ll[mask].sum())


def fit_logistic(y, X, offset, mask, p, eta, beta, opt: Step1Options, check_hs_dev: bool, numtol: float):
    """Step1_Models.cpp:156-222.  Returns (ok, beta, p, eta)."""
    dev_old = logist_dev(y, p, mask)
    niter = 0
    betanew = beta.copy()
    diff_dev = 0.0
    small_score = False
    while True:
        niter += 1
        if niter > opt.niter_max:
            break
        w, bad = get_wvec(p, mask)
        if bad:
            return False, beta, p, eta
        wm = np.where(mask, w, 0.0)
        XtW = X.T * wm[None, :]
        XtWX = XtW @ X
        z = np.where(mask, eta - offset + (y - p) / w, 0.0)
        betanew = np.linalg.lstsq(XtWX, XtW @ z, rcond=None)[0]   # colPivHouseholderQr().solve
        ok_search = False
        for _ in range(opt.niter_max_line_search):
            eta = offset + X @ betanew
            p = get_pvec(eta)
            dev_new = logist_dev(y, p, mask)
            pm = p[mask]
            if ((pm > 0) & (pm < 1)).all() and ((not check_hs_dev) or dev_new < dev_old):
                ok_search = True
                break
            betanew = (beta + betanew) / 2
        if not ok_search:
            return False, beta, p, eta
        score = X.T @ np.where(mask, y - p, 0.0)
        smax = np.abs(score).max()
        if smax < numtol:
            break
        if (not small_score) and niter < 20 and smax < 1:
            small_score = True
        if small_score and niter > 20 and smax > 5:
            return False, beta, p, eta
        diff_dev = abs(dev_new - dev_old) / (0.1 + abs(dev_new))
        beta = betanew
        dev_old = dev_new
    if (diff_dev == 0 or diff_dev >= numtol) and niter > opt.niter_max:
        return False, beta, p, eta
    return True, betanew, p, eta


def fit_null_logistic(prep: Prepared, opt: Step1Options) -> None:
    """Step1_Models.cpp:54-154 (step-1 branch): covariate-only logistic -> offset_nullreg."""
    N, P = prep.Y.shape
    prep.offset = np.zeros((N, P))
    for ph in range(P):
        y = prep.Y_raw[:, ph]
        mask = prep.mask[:, ph]
        off = np.zeros(N)
        beta0 = np.zeros(prep.X.shape[1])
        eta = off + prep.X @ beta0
        p = get_pvec(eta)
        ok, beta, p, eta = fit_logistic(y, prep.X, off, mask, p, eta, beta0, opt, True, NUMTOL)
        if not ok:      # `fit_logistic(.., true, ..) || fit_logistic(.., false, ..)` on the same pivec / etavec / betaold (:88): the second attempt goes on
            ok, beta, p, eta = fit_logistic(y, prep.X, off, mask, p, eta, beta, opt, False, NUMTOL)      # from the state the first one left
        if not ok:
            prep.pheno_pass[ph] = False
            continue
        prep.offset[:, ph] = eta


def poisson_dev(y: np.ndarray, p: np.ndarray, mask: np.ndarray) -> float:
    """get_poisson_dev / compute_log_lik_poisson (Step1_Models.cpp:1830-1849): 2 * sum -(y log p - p)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        ll = -(y * np.log(p) - p)
    return 2.0 * float(ll[mask].sum())


def fit_poisson(y, X, offset, mask, p, eta, beta, opt: Step1Options):
    """Step1_Models.cpp:290-345.  Returns (ok, beta, p, eta)."""
    dev_old = poisson_dev(y, p, mask)
    niter = 0
    dev_conv = False
    betanew = beta
    while True:
        niter += 1
        if niter > opt.niter_max:
            break
        if (p[mask] == 0).any():
            return False, beta, p, eta
        wm = np.where(mask, p, 0.0)
        XtW = X.T * wm[None, :]
        XtWX = XtW @ X
        with np.errstate(divide="ignore", invalid="ignore"):
            z = np.where(mask, eta - offset + (y - p) / p, 0.0)
        betanew = np.linalg.lstsq(XtWX, XtW @ z, rcond=None)[0]        # colPivHouseholderQr().solve
        dev_new = dev_old
        for _ in range(opt.niter_max_line_search):
            eta = offset + X @ betanew
            with np.errstate(over="ignore"):
                p = np.exp(eta)
            dev_new = poisson_dev(y, p, mask)
            if not (p[mask] == 0).any():
                break
            betanew = (beta + betanew) / 2
        score = X.T @ np.where(mask, y - p, 0.0)
        dev_conv = abs(dev_new - dev_old) / (0.1 + abs(dev_new)) < 1e-8        # params->tol
        if np.abs(score).max() < 1e-8:
            break
        beta = betanew
        dev_old = dev_new
    if (not dev_conv) and niter > opt.niter_max:
        return False, beta, p, eta
    return True, betanew, p, eta


def fit_null_poisson(prep: Prepared, opt: Step1Options) -> None:
    """Step1_Models.cpp:225-288 (step-1 branch): covariate-only Poisson regression -> offset_nullreg = eta."""
    N, P = prep.Y.shape
    prep.offset = np.zeros((N, P))
    for ph in range(P):
        y = prep.Y_raw[:, ph]
        mask = prep.mask[:, ph]
        off = np.zeros(N)
        p = y + 1e-1
        with np.errstate(divide="ignore", invalid="ignore"):
            eta = np.where(mask, np.log(p), 0.0)
        beta0 = np.zeros(prep.X.shape[1])
        beta0[0] = eta.mean() - off.mean()
        ok, beta, p, eta = fit_poisson(y, prep.X, off, mask, p, eta, beta0, opt)
        if not ok:
            prep.pheno_pass[ph] = False
            continue
        prep.offset[:, ph] = eta


def prep_run(prep: Prepared, opt: Step1Options) -> None:
    """Pheno.cpp:1060-1202 prep_run (step-1 subset): getBasis, null models, residualize_phenotypes
    (:1799-1834)."""
    prep.X, prep.ncov = get_basis(prep.X)
    if opt.bt:
        fit_null_logistic(prep, opt)
    elif opt.ct:
        fit_null_poisson(prep, opt)
    beta = prep.Y.T @ prep.X                              # P x C
    prep.Y = prep.Y - (prep.X @ beta.T) * prep.mask
    prep.scale_Y = np.linalg.norm(prep.Y, axis=0) / np.sqrt(prep.Neff - prep.ncov)
    prep.scale_Y = np.where(prep.pheno_pass, prep.scale_Y, 1.0)
    if prep.scale_Y.min() < NUMTOL:
        raise ValueError("phenotype '%s' has sd=0." % prep.pheno_names[int(np.argmin(prep.scale_Y))])
    prep.Y = prep.Y / prep.scale_Y[None, :]


# --------------------------------------------------------------------------
# folds / blocks (Data.cpp:311-398 set_blocks, :401-475 set_folds, :579-586 get_block_size)
# --------------------------------------------------------------------------
def set_folds(ind_in_analysis: np.ndarray, cv_folds: int) -> np.ndarray:
    """Data.cpp:409-426: contiguous folds in file order; each of the first K-1 folds ends at the
    sample where its count of analysed samples reaches floor(n_analyzed/K); last takes the rest."""
    N = ind_in_analysis.size
    target = int(ind_in_analysis.sum()) // cv_folds
    if target < 1:
        raise ValueError("not enough samples are present for %d-fold CV." % cv_folds)
    sizes = np.ones(cv_folds, np.int64)
    n_non_miss, cum, cur = 0, 0, 0
    for i in range(N):
        if ind_in_analysis[i]:
            n_non_miss += 1
        if n_non_miss == target:
            sizes[cur] = i - cum + 1
            cum += sizes[cur]
            n_non_miss = 0
            cur += 1
        elif cur == cv_folds - 1:
            sizes[cur] = N - i
            break
    return sizes


def chrom_blocks(chrom: np.ndarray, chr_read: List[int], bsize: int, n_block: int = 0) -> List[Tuple[int, int, int]]:
    """Data.cpp:319-333 + :579-586: list of (chrom, start_index_in_kept_snps, bs).  n_block > 0 (--nb): at most that many blocks in
    all, taken chromosome by chromosome (set_blocks, Data.cpp:314-329): the variants past them are not analysed."""
    out = []
    pos = 0
    left = n_block
    for c in chr_read:
        n = int((chrom == c).sum())
        nb = int(math.ceil(n / bsize))
        if n_block > 0:
            nb = min(nb, left)
            left -= nb
        for bb in range(nb):
            bs = bsize if (bb + 1) * bsize <= n else n - bb * bsize
            out.append((c, pos + bb * bsize, bs))
        pos += n
    return out


# --------------------------------------------------------------------------
# level 0
# --------------------------------------------------------------------------
def residualize_genotypes(G: np.ndarray, prep: Prepared) -> Tuple[np.ndarray, np.ndarray]:
    """Data.cpp:190-224.  G: bs x N (imputed).  Returns (G standardised, scale_G)."""
    G = G * prep.ind_in_analysis[None, :]
    beta = G @ prep.X
    G = G - beta @ prep.X.T
    scale_G = np.linalg.norm(G, axis=1) / math.sqrt(prep.n_analyzed - prep.ncov)
    if scale_G.min() < NUMTOL:
        j = int(np.argmin(scale_G))
        raise ValueError("!! Uh-oh, SNP #%d has low variance (=%f)." % (j, scale_G[j]))
    G = G / scale_G[:, None]
    return G, scale_G


def ridge_level_0(G: np.ndarray, prep: Prepared, cv_sizes: np.ndarray, lam: np.ndarray) -> np.ndarray:
    """Data.cpp:741-751 calc_cv_matrices (K-fold) + Step1_Models.cpp:458-613 ridge_level_0.
    Returns the block's standardised level-0 predictors, shape (P, N, R0)."""
    bs, N = G.shape
    P = prep.Y.shape[1]
    R0 = lam.size
    K = cv_sizes.size
    starts = np.concatenate([[0], np.cumsum(cv_sizes)])
    GtY, Gf = [], []
    GGt = np.zeros((bs, bs))
    GTY = np.zeros((bs, P))
    for i in range(K):
        Gi = G[:, starts[i]:starts[i + 1]]
        gty = Gi @ prep.Y[starts[i]:starts[i + 1]]
        GtY.append(gty)
        GTY += gty
        gg = Gi @ Gi.T
        Gf.append(gg)
        GGt += gg
    W = np.zeros((P, N, R0))
    p_sum = np.zeros((R0, P))
    p_sum2 = np.zeros((R0, P))
    for i in range(K):
        sl = slice(starts[i], starts[i + 1])
        d, V = np.linalg.eigh(GGt - Gf[i])
        ww2 = V.T @ (GTY - GtY[i])
        m = prep.mask[sl].T.astype(np.float64)            # P x n_i
        for j in range(R0):
            beta = V @ (ww2 / (d + lam[j])[:, None])       # bs x P
            pred = (beta.T @ G[:, sl]) * m                  # P x n_i
            p_sum[j] += pred.sum(axis=1)
            p_sum2[j] += (pred ** 2).sum(axis=1)
            W[:, sl, j] = pred
    for ph in range(P):
        mean = p_sum[:, ph] / prep.Neff[ph]
        invsd = np.sqrt((prep.Neff[ph] - 1) / (p_sum2[:, ph] - prep.Neff[ph] * mean ** 2))
        W[ph] = (W[ph] - mean[None, :]) * invsd[None, :]   # applied to ALL rows (:556-557)
    return W


def ridge_level_0_loocv(G: np.ndarray, prep: Prepared, lam: np.ndarray) -> np.ndarray:
    """Data.cpp:755-767 (LOOCV branch of calc_cv_matrices) + Step1_Models.cpp:615-726."""
    bs, N = G.shape
    P = prep.Y.shape[1]
    R0 = lam.size
    GGt = G @ G.T
    GTY = G @ prep.Y
    d, V = np.linalg.eigh(GGt)
    Wmat = V.T @ GTY                                       # bs x P
    DL_inv = 1.0 / (d[:, None] + lam[None, :])             # bs x R0
    W = np.zeros((P, N, R0))
    step = max(1, int(2e8 // max(bs, 1)))
    for s in range(0, N, step):
        e = min(N, s + step)
        VtG = V.T @ G[:, s:e]                               # bs x n_c
        gvec = (VtG ** 2).T @ DL_inv                        # n_c x R0
        for ph in range(P):
            num = VtG.T @ (DL_inv * Wmat[:, ph:ph + 1])     # n_c x R0
            W[ph, s:e] = (num - gvec * prep.Y[s:e, ph:ph + 1]) / (1 - gvec)
    for ph in range(P):
        m = prep.mask[:, ph].astype(np.float64)[:, None]
        Wp = W[ph] * m
        mean = Wp.sum(axis=0) / prep.Neff[ph]
        Wp = (Wp - mean[None, :]) * m
        sd = np.linalg.norm(Wp, axis=0) / math.sqrt(prep.Neff[ph] - 1)
        W[ph] = Wp / sd[None, :]
    return W


# --------------------------------------------------------------------------
# level 1
# --------------------------------------------------------------------------
def tau_from_h(h: np.ndarray, L: int, bt: bool) -> np.ndarray:
    """Step1_Models.cpp:2115-2117 check_l0 default branch."""
    tau = L * (1 - h) / h
    if bt:
        tau = tau * 3 / (math.pi ** 2)
    return tau


def ridge_level_1(W: np.ndarray, y: np.ndarray, cv_sizes: np.ndarray, tau: np.ndarray):
    """Step1_Models.cpp:772-872 for ONE phenotype.  W: N x L, y: N.
    Returns (cumsum[5, R1], beta_hat[K] each L x R1)."""
    N, L = W.shape
    K = cv_sizes.size
    R1 = tau.size
    starts = np.concatenate([[0], np.cumsum(cv_sizes)])
    Xf, Xy = [], []
    S = np.zeros((L, L))
    t = np.zeros(L)
    for i in range(K):
        Wi = W[starts[i]:starts[i + 1]]
        Xf.append(Wi.T @ Wi)
        Xy.append(Wi.T @ y[starts[i]:starts[i + 1]])
        S += Xf[-1]
        t += Xy[-1]
    cs = np.zeros((5, R1))
    betas = []
    for i in range(K):
        d, V = np.linalg.eigh(S - Xf[i])
        VtX2 = V.T @ (t - Xy[i])
        beta = V @ (VtX2[:, None] / (d[:, None] + tau[None, :]))      # L x R1
        betas.append(beta)
        yi = y[starts[i]:starts[i + 1]]
        p1 = W[starts[i]:starts[i + 1]] @ beta
        cs[0] += p1.sum(axis=0)
        cs[1] += yi.sum()
        cs[2] += (p1 ** 2).sum(axis=0)
        cs[3] += (yi ** 2).sum()
        cs[4] += (p1 * yi[:, None]).sum(axis=0)
    return cs, betas


def ridge_level_1_loocv(W: np.ndarray, y: np.ndarray, tau: np.ndarray, Neff: float, ncov: int):
    """Step1_Models.cpp:875-962 for ONE phenotype."""
    R1 = tau.size
    cs = np.zeros((5, R1))
    cs[3] += Neff - ncov
    xtx = W.T @ W
    d, V = np.linalg.eigh(xtx)
    zvec = V.T @ (W.T @ y)
    T = W @ V
    for j in range(R1):
        tv = 1.0 / (d + tau[j])
        cal = (T ** 2) @ tv
        pred = T @ (tv * zvec) - cal * y
        pred = pred / (1 - cal)
        cs[0, j] += pred.sum()
        cs[2, j] += pred @ pred
        cs[4, j] += pred @ y
    return cs


def _llt(A):
    """Cholesky factor, or None when A is not (numerically) positive definite or not finite.  The reference solves its Newton systems with Eigen's
    `A.llt().solve(b)` (Step1_Models.cpp:1052, :1325, :1499, :1723), which does not report a failed factorisation: the step is then garbage, the
    stopping rule is never met, and the trait ends as "did not converge" after --niter rounds.  That happens on count traits whose rows are missing
    for some traits only: the raw column keeps the missing-value code there, the rate and with it the penalties come out
    negative or NaN.  The callers below return "not converged" at once (tests/golden/fuzz_oracle_vs_reference.py draws such cases)."""
    if not np.isfinite(A).all():
        return None
    try:
        return np.linalg.cholesky(A)
    except np.linalg.LinAlgError:
        return None


def run_log_ridge_loocv(lam: float, beta: np.ndarray, y, X, offset, mask, opt: Step1Options):
    """Step1_Models.cpp:1288-1374.  Returns (ok, beta, p, w)."""
    eta = offset + X @ beta
    p = get_pvec(eta)
    fn_start = logist_dev(y, p, mask) + lam * (beta ** 2).sum()
    w, bad = get_wvec(p, mask)
    if bad:
        return False, beta, p, w
    score = X.T @ np.where(mask, y - p, 0.0) - lam * beta
    niter = 0
    dev_conv = False
    betanew = beta
    while True:
        niter += 1
        if niter > opt.niter_max_ridge:
            break
        wm = np.where(mask, w, 0.0)
        XtWX = (X.T * wm[None, :]) @ X
        XtWX[np.diag_indices_from(XtWX)] += lam
        cho = _llt(XtWX)
        if cho is None:
            return False, beta, p, w
        step = np.linalg.solve(cho.T, np.linalg.solve(cho, score))
        for _ in range(opt.niter_max_line_search):
            betanew = beta + step
            eta = offset + X @ betanew
            p = get_pvec(eta)
            fn_end = logist_dev(y, p, mask) + lam * (betanew ** 2).sum()
            w, bad = get_wvec(p, mask)
            if bad:
                return False, beta, p, w
            if fn_end < fn_start + NUMTOL:
                break
            step = step / 2
        score = X.T @ np.where(mask, y - p, 0.0) - lam * betanew
        dev_conv = abs(fn_end - fn_start) / (0.01 + abs(fn_end)) < 1e-8     # params->tol
        if np.abs(score).max() < L1_RIDGE_TOL:
            break
        beta = betanew
        fn_start = fn_end
    if (not dev_conv) and niter > opt.niter_max_ridge:
        return False, beta, p, w
    return True, betanew, p, w


def _loo_betas(X, y, p, w, mask, beta, lam):
    """Shared LOO shortcut of Step1_Models.cpp:1221-1251 / Data.cpp:1521-1548."""
    wm = np.where(mask, w, 0.0)
    XtWX = (X.T * wm[None, :]) @ X
    XtWX[np.diag_indices_from(XtWX)] += lam
    cho = np.linalg.cholesky(XtWX)
    V1 = np.linalg.solve(cho.T, np.linalg.solve(cho, X.T))        # L x N
    v2 = (X * V1.T).sum(axis=1) * w
    b_loo = beta[:, None] - V1 * ((y - p) / (1 - v2))[None, :]     # L x N
    return b_loo


def ridge_logistic_level_1_loocv(W, yraw, offset, mask, tau, opt: Step1Options):
    """Step1_Models.cpp:1159-1286 for ONE phenotype.  Returns (cumsum[6,R1], converged)."""
    N, L = W.shape
    R1 = tau.size
    cs = np.zeros((6, R1))
    beta = np.zeros(L)
    for j in range(R1):
        ok, beta, p, w = run_log_ridge_loocv(tau[j], beta, yraw, W, offset, mask, opt)
        if not ok:
            return cs, False
        b_loo = _loo_betas(W, yraw, p, w, mask, beta, tau[j])
        pred = (W * b_loo.T).sum(axis=1) + offset
        with np.errstate(over="ignore"):
            p1 = 1 - 1 / (np.exp(pred) + 1)
        p1 = np.clip(p1, L1_RIDGE_EPS, 1 - L1_RIDGE_EPS)
        m = mask
        ym = yraw[m]
        pm = p1[m]
        cs[0, j] = pm.sum()
        cs[1, j] = ym.sum()
        cs[2, j] = (pm * pm).sum()
        cs[3, j] = (ym * ym).sum()
        cs[4, j] = (pm * ym).sum()
        cs[5, j] = (-np.where(ym == 0, np.log(1 - pm), np.log(pm))).sum()
    return cs, True


def _xtwx(Xt: np.ndarray, wm: np.ndarray) -> np.ndarray:
    """X^T diag(w) X.  Small inputs: the one product the reference forms (XtW * X, Step1_Models.cpp:1047-1049).  Above 2e8 entries the same sum
    is taken over row chunks on a few threads (numpy's elementwise products are single-threaded; at 400,000 x 2,560 the 8 GB temporary of the
    one-line form costs more than the product itself) -- bench.py's full-size check of the logistic ridge is what needs it."""
    n, L = Xt.shape
    if n * L <= 2e8:
        return (Xt.T * wm[None, :]) @ Xt
    from concurrent.futures import ThreadPoolExecutor
    step = 32768
    parts = [(a, min(n, a + step)) for a in range(0, n, step)]

    def one(ab):
        a, b = ab
        Xc = Xt[a:b]
        return (Xc.T * wm[None, a:b]) @ Xc
    acc = np.zeros((L, L))
    with ThreadPoolExecutor(max_workers=8) as ex:
        for g in ex.map(one, parts):                 # results arrive in chunk order: a fixed summation order
            acc += g
    return acc


def ridge_logistic_level_1(W, yraw, offset, mask, cv_sizes, tau, opt: Step1Options, folds=None):
    """Step1_Models.cpp:966-1156 (out-of-sample branch) for ONE phenotype.
    Returns (cumsum[6,R1], betas[K] each L x R1, converged).  folds: restrict the run to these fold models (the sums are then their shares)."""
    N, L = W.shape
    K = cv_sizes.size
    R1 = tau.size
    starts = np.concatenate([[0], np.cumsum(cv_sizes)])
    cs = np.zeros((6, R1))
    betas = [np.zeros((L, R1)) for _ in range(K)]
    for i in (range(K) if folds is None else folds):
        tr = np.ones(N, bool)
        tr[starts[i]:starts[i + 1]] = False
        Xt, yt, ot, mt = W[tr], yraw[tr], offset[tr], mask[tr]
        betanew = np.zeros(L)
        for j in range(R1):
            betaold = betanew
            niter = 0
            while True:
                niter += 1
                if niter > opt.niter_max_ridge:
                    break
                eta = ot + Xt @ betaold
                p = get_pvec(eta)
                w, bad = get_wvec(p, mt)
                if bad:
                    return cs, betas, False
                z = np.where(mt, (eta - ot) + (yt - p) / w, 0.0)
                wm = np.where(mt, w, 0.0)
                XtWX = _xtwx(Xt, wm)
                XtWX[np.diag_indices_from(XtWX)] += tau[j]
                cho = _llt(XtWX)
                if cho is None:
                    return cs, betas, False
                betanew = np.linalg.solve(cho.T, np.linalg.solve(cho, Xt.T @ (wm * z)))
                for _ in range(opt.niter_max_line_search_ridge):
                    p = get_pvec(ot + Xt @ betanew)
                    w, bad = get_wvec(p, mt)
                    if not bad:
                        break
                    betanew = (betaold + betanew) / 2
                p = get_pvec(ot + Xt @ betanew)
                w, bad = get_wvec(p, mt)
                if bad:
                    return cs, betas, False
                score = Xt.T @ np.where(mt, yt - p, 0.0) - tau[j] * betanew
                if np.abs(score).max() < L1_RIDGE_TOL:
                    break
                betaold = betanew
            if niter > opt.niter_max_ridge:
                return cs, betas, False
            sl = slice(starts[i], starts[i + 1])
            etat = offset[sl] + W[sl] @ betanew
            with np.errstate(over="ignore"):
                p1 = 1 - 1 / (np.exp(etat) + 1)
            betas[i][:, j] = betanew
            m = mask[sl]
            pm = np.clip(p1[m], L1_RIDGE_EPS, 1 - L1_RIDGE_EPS)
            ym = yraw[sl][m]
            cs[0, j] += pm.sum()
            cs[1, j] += ym.sum()
            cs[2, j] += (pm * pm).sum()
            cs[3, j] += (ym * ym).sum()
            cs[4, j] += (pm * ym).sum()
            cs[5, j] += (-np.where(ym == 0, np.log(1 - pm), np.log(pm))).sum()
    return cs, betas, True


def _poisson_sums(cs, j, p1, y):
    """The six running sums of Step1_Models.cpp:1557-1570 / :1672-1684 (p1 clamped below at l1_ridge_eps)."""
    p1 = np.maximum(p1, L1_RIDGE_EPS)
    cs[0, j] += p1.sum()
    cs[1, j] += y.sum()
    cs[2, j] += (p1 * p1).sum()
    cs[3, j] += (y * y).sum()
    cs[4, j] += (p1 * y).sum()
    cs[5, j] += (-(y * np.log(p1) - p1)).sum()


def ridge_poisson_level_1(W, yraw, offset, mask, cv_sizes, tau, opt: Step1Options):
    """Step1_Models.cpp:1429-1583 for ONE phenotype.  Returns (cumsum[6,R1], betas[K] each L x R1, converged)."""
    N, L = W.shape
    K = cv_sizes.size
    R1 = tau.size
    starts = np.concatenate([[0], np.cumsum(cv_sizes)])
    cs = np.zeros((6, R1))
    betas = [np.zeros((L, R1)) for _ in range(K)]

    def pvec(b, X, o):
        with np.errstate(over="ignore"):
            e = o + X @ b
            return e, np.exp(e)

    for i in range(K):
        tr = np.ones(N, bool)
        tr[starts[i]:starts[i + 1]] = False
        Xt, yt, ot, mt = W[tr], yraw[tr], offset[tr], mask[tr]
        betanew = np.zeros(L)
        for j in range(R1):
            betaold = betanew
            niter = 0
            while True:
                niter += 1
                if niter > opt.niter_max_ridge:
                    break
                eta, p = pvec(betaold, Xt, ot)
                if (p[mt] == 0).any():
                    return cs, betas, False
                with np.errstate(divide="ignore", invalid="ignore"):
                    z = np.where(mt, (eta - ot) + (yt - p) / p, 0.0)
                wm = np.where(mt, p, 0.0)
                XtW = Xt.T * wm[None, :]
                XtWX = XtW @ Xt
                XtWX[np.diag_indices_from(XtWX)] += tau[j]
                cho = _llt(XtWX)
                if cho is None:
                    return cs, betas, False
                betanew = np.linalg.solve(cho.T, np.linalg.solve(cho, XtW @ z))
                for _ in range(opt.niter_max_line_search_ridge):
                    _, p = pvec(betanew, Xt, ot)
                    if not (p[mt] == 0).any():
                        break
                    betanew = (betaold + betanew) / 2
                _, p = pvec(betanew, Xt, ot)
                if (p[mt] == 0).any():
                    return cs, betas, False
                score = Xt.T @ np.where(mt, yt - p, 0.0) - tau[j] * betanew
                if np.abs(score).max() < L1_RIDGE_TOL:
                    break
                betaold = betanew
            if niter > opt.niter_max_ridge:
                return cs, betas, False
            sl = slice(starts[i], starts[i + 1])
            _, p1 = pvec(betanew, W[sl], offset[sl])
            betas[i][:, j] = betanew
            m = mask[sl]
            _poisson_sums(cs, j, p1[m], yraw[sl][m])
    return cs, betas, True


def run_ct_ridge_loocv(lam: float, beta: np.ndarray, y, X, offset, mask, opt: Step1Options):
    """Step1_Models.cpp:1694-1758: plain Newton (no line search).  Returns (ok, beta, p)."""
    betaold = beta
    betanew = beta
    niter = 0
    p = None
    while True:
        niter += 1
        if niter > opt.niter_max_ridge:
            break
        with np.errstate(over="ignore"):
            eta = offset + X @ betaold
            p = np.exp(eta)
        if (p[mask] == 0).any():
            return False, betaold, p
        with np.errstate(divide="ignore", invalid="ignore"):
            z = np.where(mask, (eta - offset) + (y - p) / p, 0.0)
        wm = np.where(mask, p, 0.0)
        XtW = X.T * wm[None, :]
        XtWX = XtW @ X
        XtWX[np.diag_indices_from(XtWX)] += lam
        cho = _llt(XtWX)
        if cho is None:
            return False, betaold, p
        betanew = np.linalg.solve(cho.T, np.linalg.solve(cho, XtW @ z))
        with np.errstate(over="ignore"):
            p = np.exp(offset + X @ betanew)
        if (p[mask] == 0).any():
            return False, betaold, p
        score = X.T @ np.where(mask, y - p, 0.0) - lam * betanew
        if np.abs(score).max() < L1_RIDGE_TOL:
            break
        betaold = betanew
    if niter > opt.niter_max_ridge:
        return False, betaold, p
    return True, betanew, p


def ridge_poisson_level_1_loocv(W, yraw, offset, mask, tau, opt: Step1Options):
    """Step1_Models.cpp:1585-1692 for ONE phenotype.  Returns (cumsum[6,R1], converged)."""
    N, L = W.shape
    R1 = tau.size
    cs = np.zeros((6, R1))
    beta = np.zeros(L)
    for j in range(R1):
        ok, beta, p = run_ct_ridge_loocv(tau[j], beta, yraw, W, offset, mask, opt)
        if not ok:
            return cs, False
        b_loo = _loo_betas(W, yraw, p, p, mask, beta, tau[j])            # weights of a Poisson model are its means
        with np.errstate(over="ignore"):
            p1 = np.exp((W * b_loo.T).sum(axis=1) + offset)
        _poisson_sums(cs, j, p1[mask], yraw[mask])
    return cs, True


def make_predictions_count_loocv(W, yraw, offset, mask, tau_best, chrcols, opt: Step1Options):
    """Data.cpp:1625-1712: refit at tau* from beta = 0.  Its Hessian sums the weights of ALL rows (no mask); the rows
    of masked samples in the LOOCV level-0 predictors are zero (Step1_Models.cpp:693-704), so that is the same matrix."""
    N, L = W.shape
    ok, beta, p = run_ct_ridge_loocv(tau_best, np.zeros(L), yraw, W, offset, mask, opt)
    XtWX = (W.T * p[None, :]) @ W
    XtWX[np.diag_indices_from(XtWX)] += tau_best
    cho = np.linalg.cholesky(XtWX)
    V1 = np.linalg.solve(cho.T, np.linalg.solve(cho, W.T))
    v2 = (W * V1.T).sum(axis=1) * p
    b_loo = beta[:, None] - V1 * ((yraw - p) / (1 - v2))[None, :]
    pred = np.zeros((N, len(chrcols)))
    for ci, (_, ctr, nn) in enumerate(chrcols):
        pred[:, ci] = (W[:, ctr:ctr + nn] * b_loo[ctr:ctr + nn].T).sum(axis=1)
    return pred


def tau_count(h: np.ndarray, L: int, yraw_col: np.ndarray, Neff: float) -> np.ndarray:
    """check_l0, Step1_Models.cpp:2101-2104: tau_j = L / log(1 + h_j / (rate (1 - h_j))), rate = sum(raw) / Neff.
    (The reference sums the raw column as it is: a sample kept in the analysis but missing for this phenotype
    contributes its missing-value code; restated as is.)"""
    rate = yraw_col.sum() / Neff
    return L / np.log(1 + h / (rate * (1 - h)))


# --------------------------------------------------------------------------
# output stage (Data.cpp:956-1129 output, :1196-1342, :1346-1427, :1484-1571, :1795-1975)
# --------------------------------------------------------------------------
def select_tau(cs: np.ndarray, Neff: float, bt: bool) -> int:
    """Data.cpp:1025-1037 (strict '<' keeps the first minimum)."""
    if bt:
        perf = cs[5] / Neff
    else:
        perf = (cs[2] + cs[3] - 2 * cs[4]) / Neff
    best, minv = 0, 1e10
    for j in range(perf.size):
        if perf[j] < minv:
            best, minv = j, perf[j]
    return best


def cv_table(cs: np.ndarray, Neff: float, L: int, tau: np.ndarray, bt: bool, best: int, ct_rate: Optional[float] = None) -> List[str]:
    """The per-tau log lines of Data.cpp:1042-1077 (ct_rate: count traits, :1039-1054 -- no MSE column)."""
    out = []
    for j in range(tau.size):
        if ct_rate is not None:
            zv = math.exp(L / tau[j]) - 1
            h = ct_rate * zv / (1 + ct_rate * zv)
        else:
            h = L / (L + ((math.pi ** 2 / 3) if bt else 1.0) * tau[j])
        num = cs[4, j] - cs[0, j] * cs[1, j] / Neff
        rsq = num * num / ((cs[2, j] - cs[0, j] ** 2 / Neff) * (cs[3, j] - cs[1, j] ** 2 / Neff))
        sse = cs[2, j] + cs[3, j] - 2 * cs[4, j]
        s = "  %-5s : Rsq = %s" % (cpp_double(h).rjust(5), cpp_double(rsq))
        if ct_rate is None:
            s += ", MSE = %s" % cpp_double(sse / Neff)
        if bt or ct_rate is not None:
            s += ", -logLik/N = %s" % cpp_double(cs[5, j] / Neff)
        if j == best:
            s += "<- min value"
        out.append(s)
    return out


def cpp_double(v: float) -> str:
    """Default C++ ostream formatting of a double (precision 6, %g)."""
    return "%.6g" % v


def chr_columns(blocks: List[Tuple[int, int, int]], chr_read: List[int], R0: int) -> List[Tuple[int, int, int]]:
    """(chrom, first_col, ncols) for each chromosome that has blocks (Data.cpp:1242-1249)."""
    out = []
    ctr = 0
    for c in chr_read:
        nb = sum(1 for b in blocks if b[0] == c)
        nn = nb * R0
        if nn > 0:
            out.append((c, ctr, nn))
            ctr += nn
    return out


def make_predictions(W, betas, best, cv_sizes, chrcols):
    """Data.cpp:1196-1266 (QT K-fold) and :1346-1427 (BT K-fold): out-of-fold betas."""
    N = W.shape[0]
    starts = np.concatenate([[0], np.cumsum(cv_sizes)])
    pred = np.zeros((N, len(chrcols)))
    for ci, (_, ctr, nn) in enumerate(chrcols):
        for i in range(cv_sizes.size):
            sl = slice(starts[i], starts[i + 1])
            pred[sl, ci] = W[sl, ctr:ctr + nn] @ betas[i][ctr:ctr + nn, best]
    return pred


def make_predictions_loocv(W, y, tau_best, chrcols):
    """Data.cpp:1269-1342 (QT LOOCV)."""
    N, L = W.shape
    xtx = W.T @ W
    xtx[np.diag_indices_from(xtx)] += tau_best
    zvec = W.T @ y
    d, V = np.linalg.eigh(xtx)
    H = (V / d[None, :]) @ V.T
    bvec = H @ zvec
    yres = y - W @ bvec
    HX = H @ W.T                                          # L x N
    cal = (W * HX.T).sum(axis=1)
    b0 = bvec[:, None] - HX * (yres / (1 - cal))[None, :]
    pred = np.zeros((N, len(chrcols)))
    for ci, (_, ctr, nn) in enumerate(chrcols):
        pred[:, ci] = (W[:, ctr:ctr + nn] * b0[ctr:ctr + nn].T).sum(axis=1)
    return pred


def make_predictions_binary_loocv(W, yraw, offset, mask, tau_best, chrcols, opt: Step1Options):
    """Data.cpp:1484-1571."""
    N, L = W.shape
    ok, beta, p, w = run_log_ridge_loocv(tau_best, np.zeros(L), yraw, W, offset, mask, opt)
    b_loo = _loo_betas(W, yraw, p, w, mask, beta, tau_best)
    pred = np.zeros((N, len(chrcols)))
    for ci, (_, ctr, nn) in enumerate(chrcols):
        pred[:, ci] = (W[:, ctr:ctr + nn] * b_loo[ctr:ctr + nn].T).sum(axis=1)
    return pred


def loco_from_predictions(pred: np.ndarray, chrcols, nchrom: int) -> np.ndarray:
    """Data.cpp:1846-1858: LOCO[:,c] = rowsum - pred[:,c]; absent chromosomes get the full sum."""
    N = pred.shape[0]
    tot = pred.sum(axis=1)
    out = np.repeat(tot[:, None], nchrom, axis=1)
    for ci, (c, _, _) in enumerate(chrcols):
        out[:, c - 1] -= pred[:, ci]
    return out


def write_loco(path: str, ids: List[str], ind_in_analysis: np.ndarray, mask_ph: np.ndarray, loco: np.ndarray):
    """Data.cpp:1926-1975: header + one row per chromosome, std::map (lexicographic) id order,
    default precision 6, 'NA' where masked, trailing space."""
    order = sorted(range(len(ids)), key=lambda i: ids[i])
    order = [i for i in order if ind_in_analysis[i]]
    with open(path, "w") as fh:
        fh.write("FID_IID " + "".join(ids[i] + " " for i in order) + "\n")
        for c in range(loco.shape[1]):
            fh.write("%d " % (c + 1) + "".join(
                (cpp_double(loco[i, c]) if mask_ph[i] else "NA") + " " for i in order) + "\n")


# --------------------------------------------------------------------------
# driver (Data.cpp:95-133 run_step1)
# --------------------------------------------------------------------------
@dataclass
class Step1Result:
    prep: Prepared
    blocks: List[Tuple[int, int, int]]
    chr_read: List[int]
    cv_sizes: Optional[np.ndarray]
    lam: np.ndarray
    tau: List[np.ndarray]
    W: List[np.ndarray]                 # per phenotype N x L
    cumsum: List[np.ndarray]
    best: List[int]
    predictions: List[Optional[np.ndarray]]
    loco: List[Optional[np.ndarray]]
    log: List[str] = field(default_factory=list)
    use_loocv: bool = False
    converged: List[bool] = field(default_factory=list)
    snp_ids: List[str] = field(default_factory=list)
    snp_offsets: Optional[np.ndarray] = None


def load_inputs(opt: Step1Options):
    """file_read_initialization (Data.cpp:155-180) + read_pheno_and_cov + prep_run."""
    bim = read_bim(opt.bed + ".bim", opt.nchrom)
    fam_ids = read_fam(opt.bed + ".fam")
    keep = np.ones(len(bim.ids), bool)
    if opt.extract:
        s = read_snp_files(opt.extract)
        keep &= np.array([i in s for i in bim.ids])
    if opt.exclude:
        s = read_snp_files(opt.exclude)
        keep &= np.array([i not in s for i in bim.ids])
    if not keep.any():
        raise ValueError("no variant left to include in analysis.")
    chrom = bim.chrom[keep]
    offs = bim.offset[keep]
    snp_ids = [i for i, k in zip(bim.ids, keep) if k]
    if chrom.size > 1e6 and not opt.force_step1:
        raise ValueError("it is not recommened to use more than 1M variants in step 1")
    prep = read_pheno_and_cov(opt, fam_ids)
    prep_run(prep, opt)
    return bim, chrom, offs, snp_ids, prep


def run_step1(opt: Step1Options, write_files: bool = False, keep_W: bool = True) -> Step1Result:
    bim, chrom, offs, snp_ids, prep = load_inputs(opt)
    bed, _ = open_bed(opt.bed + ".bed", prep.n_file)
    N, P = prep.Y.shape
    bt = opt.bt
    blocks = chrom_blocks(chrom, bim.chr_read, opt.bsize, opt.n_block)
    B = len(blocks)
    use_loocv = opt.loocv
    log: List[str] = []
    if bt and not use_loocv and prep.n_analyzed < 5000:         # Data.cpp:353-356
        log.append("   -WARNING: Sample size is less than 5,000 so using LOOCV instead of %d-fold CV." % opt.cv_folds)
        use_loocv = True
    def unit_params(v, name):                                    # get_unit_params (Regenie.cpp:1477-1495): sorted, unique, inside (0, 1)
        v = np.unique(np.asarray(v, np.float64))
        if ((v <= 0) | (v >= 1)).any():
            raise ValueError("must specify values for %s in (0,1)." % name)
        return v
    h0 = unit_params(opt.setl0, "--l0") if opt.setl0 is not None else set_ridge_params(opt.n_ridge_l0)
    h1 = unit_params(opt.setl1, "--l1") if opt.setl1 is not None else set_ridge_params(opt.n_ridge_l1)
    R0 = h0.size
    M = opt.parallel_nGeno if opt.parallel_nGeno is not None else chrom.size
    lam = M * (1 - h0) / h0                                       # Data.cpp:607
    cv_sizes = None if use_loocv else set_folds(prep.ind_in_analysis, opt.cv_folds)
    if opt.ct and not use_loocv:                                  # Data.cpp:436-466: a fold without any count
        st = np.concatenate([[0], np.cumsum(cv_sizes)])
        for i in range(cv_sizes.size):
            ssum = (prep.Y_raw[st[i]:st[i + 1]] * prep.mask[st[i]:st[i + 1]]).sum(axis=0)
            ssum = np.where(prep.pheno_pass, ssum, 10.0)
            if ssum.min() == 0:
                raise ValueError("one of the folds has only zero counts for phenotype '%s'. Either use smaller #folds (option --cv) "
                                 "or use LOOCV (option --loocv)." % prep.pheno_names[int(np.argmin(ssum))])
    L = B * R0
    W = [np.zeros((N, L)) for _ in range(P)]
    for b, (c, start, bs) in enumerate(blocks):
        if opt.dosage_provider is not None:
            G = read_chunk_from_dosages(opt.dosage_provider(offs[start:start + bs]), prep.ind_ignore, prep.ind_in_analysis)
        else:
            rows = np.asarray(bed[offs[start:start + bs]])
            G = read_chunk_from_bed(rows, prep.n_file, prep.ind_ignore, prep.ind_in_analysis, opt.ref_first)
        G, _ = residualize_genotypes(G, prep)
        Wb = ridge_level_0_loocv(G, prep, lam) if use_loocv else ridge_level_0(G, prep, cv_sizes, lam)
        for ph in range(P):
            W[ph][:, b * R0:(b + 1) * R0] = Wb[ph]
    res = finish_level_1(opt, prep, blocks, bim.chr_read, cv_sizes, lam, h1, W, use_loocv, log, write_files)
    res.snp_ids = snp_ids
    res.snp_offsets = offs
    if not keep_W:
        res.W = []
    return res


def finish_level_1(opt, prep, blocks, chr_read, cv_sizes, lam, h1, W, use_loocv, log, write_files=False) -> Step1Result:
    """prep_l1_models + level-1 dispatch + output (Data.cpp:113-131)."""
    bt = opt.bt
    ct = opt.ct
    P = prep.Y.shape[1]
    R0 = lam.size
    L = len(blocks) * R0
    chrcols = chr_columns(blocks, chr_read, R0)
    taus, css, bests, preds, locos, conv = [], [], [], [], [], []
    pred_list = []
    for ph in range(P):
        tau = tau_count(h1, L, prep.Y_raw[:, ph], prep.Neff[ph]) if ct else tau_from_h(h1, L, bt)
        taus.append(tau)
        y = prep.Y[:, ph]
        betas = None
        ok = True
        if not prep.pheno_pass[ph]:
            css.append(np.zeros((6, h1.size))); bests.append(0); preds.append(None); locos.append(None); conv.append(False)
            continue
        if ct:
            if use_loocv:
                cs, ok = ridge_poisson_level_1_loocv(W[ph], prep.Y_raw[:, ph], prep.offset[:, ph], prep.mask[:, ph], tau, opt)
            else:
                cs, betas, ok = ridge_poisson_level_1(W[ph], prep.Y_raw[:, ph], prep.offset[:, ph], prep.mask[:, ph], cv_sizes, tau, opt)
        elif not bt:
            if use_loocv:
                cs = ridge_level_1_loocv(W[ph], y, tau, prep.Neff[ph], prep.ncov)
            else:
                cs, betas = ridge_level_1(W[ph], y, cv_sizes, tau)
        else:
            if use_loocv:
                cs, ok = ridge_logistic_level_1_loocv(W[ph], prep.Y_raw[:, ph], prep.offset[:, ph], prep.mask[:, ph], tau, opt)
            else:
                cs, betas, ok = ridge_logistic_level_1(W[ph], prep.Y_raw[:, ph], prep.offset[:, ph], prep.mask[:, ph], cv_sizes, tau, opt)
        css.append(cs)
        conv.append(ok)
        log.append("phenotype %d (%s) : " % (ph + 1, prep.pheno_names[ph]))
        if not ok:
            log.append("Level 1 model did not converge. LOCO predictions calculations are skipped.")
            bests.append(0); preds.append(None); locos.append(None)
            continue
        best = select_tau(cs, prep.Neff[ph], bt or ct)
        bests.append(best)
        log.extend(cv_table(cs, prep.Neff[ph], L, tau, bt, best, ct_rate=(prep.Y_raw[:, ph].sum() / prep.Neff[ph]) if ct else None))
        if ct:
            pred = make_predictions_count_loocv(W[ph], prep.Y_raw[:, ph], prep.offset[:, ph], prep.mask[:, ph], tau[best], chrcols, opt) \
                if use_loocv else make_predictions(W[ph], betas, best, cv_sizes, chrcols)    # make_predictions_count, Data.cpp:1575-1622
        elif not bt:
            pred = make_predictions_loocv(W[ph], y, tau[best], chrcols) if use_loocv else \
                make_predictions(W[ph], betas, best, cv_sizes, chrcols)
        else:
            pred = make_predictions_binary_loocv(W[ph], prep.Y_raw[:, ph], prep.offset[:, ph], prep.mask[:, ph], tau[best], chrcols, opt) \
                if use_loocv else make_predictions(W[ph], betas, best, cv_sizes, chrcols)
        preds.append(pred)
        loco = loco_from_predictions(pred, chrcols, opt.nchrom)
        locos.append(loco)
        if write_files:
            fn = "%s_%d.loco" % (opt.out, ph + 1)
            write_loco(fn, prep.ids, prep.ind_in_analysis, prep.mask[:, ph], loco)
            pred_list.append("%s %s" % (prep.pheno_names[ph], fn if opt.use_rel_path else os.path.abspath(fn)))
    if write_files:
        with open(opt.out + "_pred.list", "w") as fh:
            fh.write("".join(s + "\n" for s in pred_list))
        with open(opt.out + ".log", "w") as fh:
            fh.write("\n".join(log) + "\n")
    return Step1Result(prep=prep, blocks=blocks, chr_read=chr_read, cv_sizes=cv_sizes, lam=lam, tau=taus,
                       W=W, cumsum=css, best=bests, predictions=preds, loco=locos, log=log,
                       use_loocv=use_loocv, converged=conv)
