"""TEST INFRASTRUCTURE -- not part of the product path (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import anything under oracle/).

CPU restatement (numpy) of regenie's `--step 1 --t2e`: time-to-event phenotypes, Cox ridge at level 1.  Pinned against the outputs of
regenie itself (oracle/_ref/regenie, built from /root/reference by oracle/Makefile) by tests/test_reference_pin.py on the fixtures
tests/golden/ref_outputs/t2e_* (generator: tests/golden/make_ref_outputs.py).

What differs from the quantitative-trait run, with the reference lines each function follows:
  * phenotype file: a time column and an event column (0 / 1 / NA) per trait, --phenoColList / --eventColList (Regenie.cpp:570-587,
    Pheno.cpp:230-283); both columns are "phenotypes" of the run (n_pheno = 2 x traits), the event columns never pass to level 1
    (Pheno.cpp:1957-1965);
  * covariates: constant columns (the intercept) are dropped, the rest centred and scaled before the orthonormal basis
    (Pheno.cpp:1078-1103, getBasis :1660-1681);
  * null model: Cox regression on the covariates by cyclic coordinate descent, its linear predictor is the level-1 offset
    (fit_null_cox, Step1_Models.cpp:353-440; cox_ridge.cpp);
  * level 0: the K-fold ridge of the (residualised, scaled) time column, predictions kept as one N x L matrix (Step1_Models.cpp:760-768);
  * level 1: per fold a path of Cox ridge fits over five penalties tau_max * 10^(-6 j / 4), tau_max from the score at beta = 0; the
    deviance of each fit on the held-out fold is summed and the smallest total wins (ridge_cox_level_1, Step1_Models.cpp:2228-2305;
    cox_ridge_path, cox_ridge.cpp:204-302; survival_data.cpp);
  * predictions: out-of-fold W beta per chromosome, LOCO files as for the other traits (make_predictions_cox, Data.cpp:1714-1755).
Restated too (round 4, pinned against regenie's runs with them): --t2e-l1-pi6, and --t2e-event-l0 as what it is for a run that keeps its
level-0 predictors in memory: nothing (it only selects the level-0 FILE of the --lowmem / --run-l1 modes).  Not restated: the Newton fall-back of the null model (cox_firth.cpp) when the coordinate descent does not
converge (the oracle raises instead)."""
import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import regenie_step1 as s1

MISSING = s1.MISSING
NUMTOL_COX = 2.5e-4          # Regenie.hpp:221
CONST_COV_COX_TOL = 1e-6     # Regenie.hpp:228
L1_RIDGE_TOL = 1e-4          # Regenie.hpp:289


# --------------------------------------------------------------------------
# survival_data.cpp
# --------------------------------------------------------------------------
class SurvivalData:
    """survival_data::setup (survival_data.cpp:9-100), without the risk-set matrix R / rskcount / ww_k that only cox_firth uses."""

    def __init__(self, event_time: np.ndarray, event_status: np.ndarray, mask: np.ndarray, norm_weights: bool = True):
        n = event_time.size
        self.n = n
        self.neff = int(mask.sum())
        status = np.where(mask, event_status, -999.0)
        # _getOrder (:104-127): by time, events before censored samples at equal times.  (std::sort leaves the order inside equal keys
        # open; every quantity below is the same for any such order.)
        self.order = np.lexsort((-status, event_time))
        time_order = np.where(mask, event_time, -999.0)[self.order]
        self.status_order = status[self.order]
        self.keep_sample_order = mask[self.order]
        w = np.ones(n)
        if norm_weights:
            w = w / self.neff
        self.w_orig = np.where(mask, w, 0.0)
        self.w = np.where(self.keep_sample_order, w, 0.0)
        ev = np.flatnonzero(self.status_order == 1)
        self.n_events = ev.size
        # _findTies (:129-150): event times that occur more than once
        self.dd = np.where(self.keep_sample_order, self.status_order, 0.0)
        self.ww = self.w.copy()
        self.unique_counts: List[int] = []
        a = 0
        while a < ev.size:
            b = a + 1
            while b < ev.size and time_order[ev[b]] == time_order[ev[a]]:
                b += 1
            self.unique_counts.append(b - a)
            if b - a > 1:
                self.dd[ev[a:b]] = 0
                self.ww[ev[a:b]] = 0
                self.dd[ev[a]] = 1
                self.ww[ev[a]] = (b - a) / self.neff if norm_weights else float(b - a)
            a = b

    def permute(self, v: np.ndarray) -> np.ndarray:          # permute_mtx * v
        return v[self.order]

    def unpermute(self, v: np.ndarray) -> np.ndarray:        # permute_mtx^T * v
        out = np.empty_like(v)
        out[self.order] = v
        return out


def _revcumsum(x: np.ndarray) -> np.ndarray:
    return np.cumsum(x[::-1])[::-1]


# --------------------------------------------------------------------------
# cox_ridge.cpp
# --------------------------------------------------------------------------
class CoxRidge:
    def __init__(self, sd: SurvivalData, X: np.ndarray, offset: np.ndarray, mask: np.ndarray, lam: float, max_iter: int, max_inner: int,
                 tol: float, beta_init: Optional[np.ndarray] = None, null_deviance: float = -999.0):
        """cox_ridge::cox_ridge (cox_ridge.cpp:8-34) / reset (:36-58)."""
        self.converge = False
        self.beta = np.zeros(X.shape[1]) if beta_init is None else np.array(beta_init, dtype=np.float64)
        self.lam = lam
        self.niter, self.mxitnr, self.tol = max_iter, max_inner, tol
        self.eta = np.where(mask, X @ self.beta + offset, 0.0)
        self.eta_order = sd.permute(self.eta)
        d0 = self.cox_deviance(sd) if null_deviance == -999.0 else null_deviance
        self.deviance = [d0]
        self.objective = [d0 + lam * (self.beta ** 2).sum() / 2]
        self.gradient = None
        self.diag_hessian = None

    def cox_grad(self, sd: SurvivalData) -> None:
        """cox_ridge::coxGrad (:60-82)."""
        mean_eta = (self.eta * sd.w_orig).sum() / sd.w_orig.sum()
        exp_eta = np.exp(self.eta_order - mean_eta)
        rskden = _revcumsum(sd.w * exp_eta)
        with np.errstate(divide="ignore", invalid="ignore"):
            ww_rsk = sd.ww / rskden
            ww_rsk2 = sd.ww / rskden ** 2
        ev = sd.dd != 0
        rskdeninv_n = np.cumsum(np.where(ev, ww_rsk, 0.0))
        rskdeninv2_n = np.cumsum(np.where(ev, ww_rsk2, 0.0))
        g = sd.w * (sd.status_order - exp_eta * rskdeninv_n)
        h = (sd.w * exp_eta) ** 2 * rskdeninv2_n - sd.w * exp_eta * rskdeninv_n
        self.gradient = sd.unpermute(g)
        self.diag_hessian = sd.unpermute(h)

    def cox_loglik(self, sd: SurvivalData) -> float:
        """_coxLoglik (:84-91)."""
        rsk = _revcumsum(sd.w * np.exp(self.eta_order))
        with np.errstate(divide="ignore"):
            lg = np.where(sd.keep_sample_order, np.log(rsk), 0.0)
        log_terms = (sd.ww * lg * (sd.dd == 1)).sum()
        return float((sd.w * self.eta_order * (sd.status_order == 1)).sum() - log_terms)

    def cox_deviance(self, sd: SurvivalData) -> float:
        """_coxDeviance (:93-114)."""
        if len(sd.unique_counts) == sd.n_events:
            w_sub = np.ones(sd.n_events) / sd.neff
        else:
            w_sub = np.array(sd.unique_counts, dtype=np.float64) / sd.neff
        lsat = -(w_sub * np.log(w_sub)).sum()
        return 2 * (lsat - self.cox_loglik(sd))

    def fit(self, sd: SurvivalData, X: np.ndarray, offset: np.ndarray, mask: np.ndarray) -> None:
        """cox_ridge::fit (:116-178): IRLS with the diagonal of the Hessian, one cyclic pass over the coordinates per iteration."""
        p = X.shape[1]
        for t in range(1, self.niter + 1):
            beta_old = self.beta.copy()
            self.cox_grad(sd)
            dh = self.diag_hessian
            with np.errstate(divide="ignore", invalid="ignore"):
                z = np.where(dh != 0, self.gradient / dh, 0.0)
            z = np.where(mask, self.eta - offset, 0.0) - z
            for k in range(p):
                r = dh * (z - self.eta + offset)
                xk = X[:, k]
                self.eta = self.eta - np.where(mask, xk * self.beta[k], 0.0)
                s2 = (xk ** 2 * dh).sum()
                self.beta[k] = (r @ xk + self.beta[k] * s2) / (s2 - self.lam)
                self.eta = self.eta + np.where(mask, xk * self.beta[k], 0.0)
            self.eta_order = sd.permute(self.eta)
            dev = self.cox_deviance(sd)
            obj = dev + self.lam * (self.beta ** 2).sum() / 2
            if dev - self.deviance[t - 1] > self.tol:
                ii = 0
                while dev - self.deviance[t - 1] > self.tol:
                    ii += 1
                    if ii > self.mxitnr:
                        self.deviance.append(dev)
                        self.objective.append(obj)
                        return
                    self.beta = (self.beta + beta_old) / 2
                    self.eta = np.where(mask, X @ self.beta + offset, 0.0)
                    self.eta_order = sd.permute(self.eta)
                    dev = self.cox_deviance(sd)
                    obj = dev + self.lam * (self.beta ** 2).sum() / 2
            self.deviance.append(dev)
            self.objective.append(obj)
            score = np.abs(self.gradient @ X - self.lam * self.beta).max() if p else 0.0
            if abs(obj - self.objective[t - 1]) / (0.1 + abs(obj)) < self.tol or score < self.tol:
                self.converge = True
                break

    def null_deviance(self) -> float:
        return self.deviance[0]


def cox_ridge_path(sd: SurvivalData, X: np.ndarray, offset: np.ndarray, mask: np.ndarray, lambdas: np.ndarray, max_iter: int, max_inner: int,
                   tol: float) -> Tuple[np.ndarray, np.ndarray]:
    """cox_ridge_path with user-defined penalties (cox_ridge.cpp:204-302): largest first, every fit started from the previous solution
    and its objective measured from the previous fit's starting deviance.  Returns (beta_mx p x n_lambda, converged flags)."""
    lam = np.sort(np.asarray(lambdas, np.float64))[::-1]
    p = X.shape[1]
    beta_mx = np.zeros((p, lam.size))
    conv = np.zeros(lam.size, bool)
    fitobj = CoxRidge(sd, X, offset, mask, lam[0], max_iter, max_inner, tol)
    for k in range(lam.size):
        if k > 0:
            fitobj = CoxRidge(sd, X, offset, mask, lam[k], max_iter, max_inner, tol, beta_init=beta_old, null_deviance=nulldev_old)
        fitobj.fit(sd, X, offset, mask)
        conv[k] = fitobj.converge
        beta_old = fitobj.beta.copy()
        nulldev_old = fitobj.null_deviance()
        beta_mx[:, k] = fitobj.beta
    return beta_mx, conv


# --------------------------------------------------------------------------
# phenotypes / covariates / null model
# --------------------------------------------------------------------------
def read_t2e(opt: s1.Step1Options, t2e_map: Dict[str, str], fam_ids: List[str]) -> s1.Prepared:
    """read_pheno_and_cov for --t2e.  t2e_map: time column -> event column (files->t2e_map, a std::map: traits are visited in the
    lexicographic order of their time names).  The columns of the run are those of --phenoColList and --eventColList in FILE order."""
    sel = set(t2e_map) | set(t2e_map.values())
    o2 = s1.Step1Options(**{**opt.__dict__, "pheno_cols": tuple(sel), "bt": False, "ct": False})
    # the generic reader gives the kept samples, the covariates and the id bookkeeping; the phenotype columns are re-read below with the
    # time-to-event rules (Pheno.cpp:230-283)
    prep = s1.read_pheno_and_cov(o2, fam_ids)
    names = prep.pheno_names
    idx = {k: i for i, k in enumerate(prep.ids)}
    N, P = len(prep.ids), len(names)
    with s1._open_text(opt.pheno_file) as fh:
        lines = fh.read().splitlines()
    hdr = lines[0].rstrip("\r").split()
    col = {h: j for j, h in enumerate(hdr)}
    Y = np.zeros((N, P))
    Yraw = np.zeros((N, P))
    mask = np.ones((N, P), bool)
    in_pheno = np.zeros(N, bool)
    for line in lines[1:]:
        t = line.split()
        if not t:
            continue
        k = t[0] + "_" + t[1]
        if k not in idx:
            continue
        i = idx[k]
        in_pheno[i] = True
        all_miss = True
        for tn, en in t2e_map.items():
            ti, ei = names.index(tn), names.index(en)
            tv = s1.convert_double(t[col[tn]])
            evv = s1.convert_double(t[col[en]])
            if opt.cc12 and evv != MISSING:
                evv -= 1
            Y[i, ti] = Yraw[i, ti] = tv
            Y[i, ei] = Yraw[i, ei] = evv
            if tv < 0 and tv != MISSING:
                raise ValueError("a phenotype time value is <0 for individual: FID=%s IID=%s" % (t[0], t[1]))
            if evv not in (0, 1, MISSING):
                raise ValueError("a phenotype censor value is invalid for individual: FID=%s IID=%s" % (t[0], t[1]))
            if tv != MISSING and evv == MISSING:
                raise ValueError("a phenotype has missing censor with non-missing time for individual: FID=%s IID=%s" % (t[0], t[1]))
            if tv == MISSING:
                mask[i, ti] = mask[i, ei] = False
                Yraw[i, ei] = MISSING
            else:
                all_miss = False
        if all_miss:
            in_pheno[i] = False
    mask &= in_pheno[:, None]
    if (mask.sum(axis=0) == 0).any():
        raise ValueError("all individuals have missing/invalid values for a phenotype")
    # the covariates as the generic reader left them, un-masked by ITS idea of the analysed samples: redo Pheno.cpp:101 + setMasks
    in_cov = (prep.X != 0).any(axis=1) | prep.ind_in_analysis        # rows with covariate data (the intercept column is 1 there)
    ain = in_pheno & in_cov
    ain &= mask.any(axis=1)
    mask &= ain[:, None]
    # the generic reader already multiplied X by its own analysed-sample vector; samples it dropped for a missing TIME entry are the
    # same samples dropped here (a missing time is "NA" in both readings), so its X is the X of this run
    X = prep.X * ain[:, None]
    Y = Y * ain[:, None]
    Yraw = Yraw * ain[:, None]
    Neff = mask.sum(axis=0).astype(np.float64)
    for j in range(P):                                                 # pheno_impute_miss, trait_mode != 0 (Pheno.cpp:1924-1927)
        tot = Y[mask[:, j], j].sum() / mask[:, j].sum()
        Y[:, j] = np.where(mask[:, j], Y[:, j], tot)
    Y = Y * mask
    # Pheno.cpp:1957-1965: the event columns do not pass unless --t2e-event-l0 keeps them for level 0 (their residuals are then scaled
    # like any other column's; ridge_cox_level_1 drops them afterwards, Step1_Models.cpp:2255)
    pheno_pass = np.array([(n in t2e_map) or bool(opt.t2e_event_l0) for n in names])
    return s1.Prepared(ids=prep.ids, n_file=prep.n_file, ind_ignore=prep.ind_ignore, ind_in_analysis=ain, pheno_names=names, Y=Y, Y_raw=Yraw,
                       mask=mask, X=X, Neff=Neff, scale_Y=np.ones(P), ncov=X.shape[1], n_analyzed=int(ain.sum()), pheno_pass=pheno_pass)


def prep_run_t2e(prep: s1.Prepared, t2e_map: Dict[str, str], opt: s1.Step1Options) -> None:
    """prep_run for --t2e (Pheno.cpp:1060-1202): covariate screening + basis, null Cox model, residualize_phenotypes."""
    X = prep.X
    mu = X.mean(axis=0)
    sds = np.linalg.norm(X - mu[None, :], axis=0) / math.sqrt(prep.n_analyzed)
    keep = sds > CONST_COV_COX_TOL
    X, sds = X[:, keep], sds[keep]
    if X.shape[1] > 0:                                               # getBasis with trait_mode 3 (:1663-1667): centre ALL rows, scale
        X = (X - X.mean(axis=0)[None, :]) / sds[None, :]
        X, ncov = s1.get_basis(X)
    else:
        ncov = 0
    prep.X, prep.ncov = X, ncov
    N, P = prep.Y.shape
    prep.offset = np.zeros((N, P))
    for tn, en in t2e_map.items():                                   # fit_null_cox (Step1_Models.cpp:353-440), step 1
        ti, ei = prep.pheno_names.index(tn), prep.pheno_names.index(en)
        m = prep.mask[:, ti]
        sd = SurvivalData(prep.Y_raw[:, ti], prep.Y_raw[:, ei], m, True)
        fit = CoxRidge(sd, X, np.zeros(N), m, 0.0, opt.niter_max, opt.niter_max_line_search, NUMTOL_COX)
        fit.fit(sd, X, np.zeros(N), m)
        if not fit.converge:
            raise NotImplementedError("null Cox model: the coordinate descent did not converge (the reference falls back on cox_firth)")
        prep.offset[:, ti] = fit.eta
    beta = prep.Y.T @ X                                               # residualize_phenotypes (Pheno.cpp:1799-1834)
    prep.Y = prep.Y - (X @ beta.T) * prep.mask
    prep.scale_Y = np.linalg.norm(prep.Y, axis=0) / np.sqrt(prep.Neff - ncov)
    prep.scale_Y = np.where(prep.pheno_pass, prep.scale_Y, 1.0)
    if prep.scale_Y.min() < s1.NUMTOL:
        raise ValueError("phenotype '%s' has sd=0." % prep.pheno_names[int(np.argmin(prep.scale_Y))])
    prep.Y = prep.Y / prep.scale_Y[None, :]


# --------------------------------------------------------------------------
# level 1
# --------------------------------------------------------------------------
def ridge_cox_level_1(W: np.ndarray, time: np.ndarray, event: np.ndarray, offset: np.ndarray, mask: np.ndarray, cv_sizes: np.ndarray,
                      opt: s1.Step1Options, n_ridge_l1: int = 5, tau_in: np.ndarray = None):
    """ridge_cox_level_1 for one trait (Step1_Models.cpp:2241-2298).  Returns (tau, summed held-out deviances, per-fold beta matrices,
    converged)."""
    N = W.shape[0]
    sd0 = SurvivalData(time, event, mask, True)
    f0 = CoxRidge(sd0, W, offset, mask, 0.0, opt.niter_max, opt.niter_max_line_search, NUMTOL_COX)
    f0.cox_grad(sd0)
    lam_max = np.abs(W.T @ f0.gradient).max() / 1e-3                 # getCoxLambdaMax (:446-450)
    idx = np.linspace(0, n_ridge_l1 - 1, n_ridge_l1)
    tau = np.exp(idx / (n_ridge_l1 - 1) * math.log(1e-6) + math.log(lam_max))      # check_l0 (:2105-2113)
    if tau_in is not None:                                           # --t2e-l1-pi6: the caller's penalties (:2106-2110)
        tau = np.asarray(tau_in, np.float64)
    starts = np.concatenate([[0], np.cumsum(cv_sizes)])
    fold_id = np.zeros(N, np.int64)
    for i in range(cv_sizes.size):
        fold_id[starts[i]:starts[i + 1]] = i
    dev = np.zeros(n_ridge_l1)
    betas = []
    ok = True
    for i in range(cv_sizes.size):
        train = (fold_id != i) & mask
        test = (fold_id == i) & mask
        sdf = SurvivalData(time, event, train, True)
        beta_mx, conv = cox_ridge_path(sdf, W, offset, train, tau, opt.niter_max_ridge, opt.niter_max_line_search_ridge, L1_RIDGE_TOL)
        ok = ok and bool(conv.all())
        betas.append(beta_mx)
        sdt = SurvivalData(time, event, test, True)
        for l in range(n_ridge_l1):
            dev[l] += CoxRidge(sdt, W, offset, test, tau[l], opt.niter_max_ridge, opt.niter_max_line_search_ridge, L1_RIDGE_TOL,
                               beta_init=beta_mx[:, l]).null_deviance()
    return tau, dev, betas, ok


def run_step1_t2e(opt: s1.Step1Options, t2e_map: Dict[str, str], write_files: bool = False):
    """Data::run_step1 for --t2e.  Returns a dict: prep, W (per column of the run), per trait tau / deviance / best / loco."""
    bim = s1.read_bim(opt.bed + ".bim", opt.nchrom)
    fam_ids = s1.read_fam(opt.bed + ".fam")
    keep = np.ones(len(bim.ids), bool)
    if opt.extract:
        s = s1.read_snp_files(opt.extract)
        keep &= np.array([i in s for i in bim.ids])
    if opt.exclude:
        s = s1.read_snp_files(opt.exclude)
        keep &= np.array([i not in s for i in bim.ids])
    chrom, offs = bim.chrom[keep], bim.offset[keep]
    prep = read_t2e(opt, t2e_map, fam_ids)
    prep_run_t2e(prep, t2e_map, opt)
    bed, _ = s1.open_bed(opt.bed + ".bed", prep.n_file)
    N, P = prep.Y.shape
    blocks = s1.chrom_blocks(chrom, bim.chr_read, opt.bsize)
    h0 = np.asarray(opt.setl0, np.float64) if opt.setl0 is not None else s1.set_ridge_params(opt.n_ridge_l0)
    R0 = h0.size
    lam = chrom.size * (1 - h0) / h0
    cv_sizes = s1.set_folds(prep.ind_in_analysis, opt.cv_folds)
    L = len(blocks) * R0
    W = [np.zeros((N, L)) for _ in range(P)]
    for b, (c, start, bs) in enumerate(blocks):
        rows = np.asarray(bed[offs[start:start + bs]])
        G = s1.read_chunk_from_bed(rows, prep.n_file, prep.ind_ignore, prep.ind_in_analysis, opt.ref_first)
        G, _ = s1.residualize_genotypes(G, prep)
        Wb = s1.ridge_level_0(G, prep, cv_sizes, lam)
        for ph in range(P):
            W[ph][:, b * R0:(b + 1) * R0] = Wb[ph]
    chrcols = s1.chr_columns(blocks, bim.chr_read, R0)
    out = {"prep": prep, "W": W, "cv_sizes": cv_sizes, "traits": {}, "log": [], "pred_list": []}
    for tn in sorted(t2e_map):                                       # std::map order
        ti, ei = prep.pheno_names.index(tn), prep.pheno_names.index(t2e_map[tn])
        # --t2e-event-l0 sets l0_idx = the event column (Step1_Models.cpp:2259), but l0_idx is only the index of the level-0 FILE that
        # read_l0 loads in the --lowmem / --run-l1 modes (:2260-2261); the in-memory run fits on test_mat_conc[ph_eff] with ph_eff = the
        # time column whatever the switch says (:2258, :2269-2282) -- regenie's own outputs with and without it are byte-identical here
        # (tests/golden/ref_outputs/t2e_kfold_synth_event_l0).  This restatement is the in-memory run.
        li = ti
        tau_in = None
        if opt.t2e_l1_pi6:                                           # check_l0 (:2106-2110): L (1 - h) / h x 6 / pi^2 over the level-1 grid of h
            h1 = np.asarray(opt.setl1, np.float64) if getattr(opt, "setl1", None) is not None else s1.set_ridge_params(opt.n_ridge_l1)
            tau_in = L * (1 - h1) / h1 * 6.0 / (math.pi * math.pi)
        tau, dev, betas, ok = ridge_cox_level_1(W[li], prep.Y_raw[:, ti], prep.Y_raw[:, ei], prep.offset[:, ti], prep.mask[:, ti], cv_sizes, opt,
                                                opt.n_ridge_l1, tau_in)
        out["traits"][tn] = {"index": ti, "l0_index": li, "tau": tau, "deviance": dev, "betas": betas, "converged": ok}
    for ti, tn in enumerate(prep.pheno_names):                       # Data::output (Data.cpp:966-1110): by column of the run
        if tn not in t2e_map:
            continue
        tr = out["traits"][tn]
        out["log"].append("phenotype %d (%s) : " % (ti + 1, tn))
        if not tr["converged"]:
            out["log"].append("Level 1 model did not converge. LOCO predictions calculations are skipped.")
            continue
        best = int(np.argmin(tr["deviance"]))                        # first minimum, no division by Neff (:1031)
        tr["best"] = best
        for j in range(opt.n_ridge_l1):
            out["log"].append(" %5s : Deviance = %s%s" % (s1.cpp_double(tr["tau"][j]), s1.cpp_double(tr["deviance"][j]),
                                                         "<- min value" if j == best else ""))
        pred = s1.make_predictions(W[tr["l0_index"]], tr["betas"], best, cv_sizes, chrcols)        # make_predictions_cox (Data.cpp:1714-1755)
        tr["loco"] = s1.loco_from_predictions(pred, chrcols, opt.nchrom)
        if write_files:
            fn = "%s_%d.loco" % (opt.out, ti + 1)
            s1.write_loco(fn, prep.ids, prep.ind_in_analysis, prep.mask[:, ti], tr["loco"])
            out["pred_list"].append("%s %s" % (tn, fn if opt.use_rel_path else os.path.abspath(fn)))
    if write_files:
        with open(opt.out + "_pred.list", "w") as fh:
            fh.write("".join(s + "\n" for s in out["pred_list"]))
    return out
