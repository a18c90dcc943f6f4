"""Shared helpers of the parity tests.  The oracle is used here as the CHECKER only: it prepares the
inputs both sides consume (phenotype/covariate prep is host-side prerequisite work, SURVEY.md 8a row
a23) and produces the expected outputs; everything under test goes through the C ABI."""
from __future__ import annotations

import numpy as np

from oracle import regenie_step1 as orc
from regenie_amd.engine import Step1Engine, loco_from_predictions


def gpu_step1(opt: orc.Step1Options, nblk_env=None):
    """Runs level 0 + level 1 (QT, K-fold) through librg_step1_hip.so on the inputs the oracle prepared.
    Returns dict(W=[P][N,L], cumsum, best, pred, loco, prep, blocks, ...)."""
    bim, chrom, offs, snp_ids, prep = orc.load_inputs(opt)
    bed, bpr = orc.open_bed(opt.bed + ".bed", prep.n_file)
    blocks = orc.chrom_blocks(chrom, bim.chr_read, opt.bsize)
    B = len(blocks)
    h0 = orc.set_ridge_params(opt.n_ridge_l0)
    h1 = orc.set_ridge_params(opt.n_ridge_l1)
    M = chrom.size
    lam = M * (1 - h0) / h0
    cv_sizes = orc.set_folds(prep.ind_in_analysis, opt.cv_folds)
    eng = Step1Engine(0)
    eng.set_problem(X=prep.X, Y=prep.Y, mask=prep.mask, ind_in_analysis=prep.ind_in_analysis,
                    cv_sizes=cv_sizes, lam=lam, neff=prep.Neff, n_file=prep.n_file, n_blocks_total=B,
                    max_block_size=opt.bsize, ind_ignore=prep.ind_ignore if prep.ind_ignore.any() else None,
                    ref_first=opt.ref_first)
    rows = [np.ascontiguousarray(bed[offs[s:s + bs]]) for (_, s, bs) in blocks]
    eng.l0_blocks_host(list(range(B)), rows)
    eng.sync()
    N, P = prep.Y.shape
    R0 = lam.size
    W = [np.zeros((N, B * R0)) for _ in range(P)]
    for b in range(B):
        for ph in range(P):
            W[ph][:, b * R0:(b + 1) * R0] = eng.get_w(b, ph)
    L = B * R0
    tau = np.stack([orc.tau_from_h(h1, L, False) for _ in range(P)])
    chrcols = orc.chr_columns(blocks, bim.chr_read, R0)
    cs, best, pred = eng.l1_qt(tau, [nn for (_, _, nn) in chrcols])
    loco = [loco_from_predictions(pred[ph], [c for (c, _, _) in chrcols], opt.nchrom) for ph in range(P)]
    eng.close()
    return dict(W=W, cumsum=cs, best=best, pred=pred, loco=loco, prep=prep, blocks=blocks, lam=lam,
                cv_sizes=cv_sizes, tau=tau, chrcols=chrcols, rows=rows)


def rel_err(a, b):
    """BASELINE.json's accuracy metric: max|a-b| / max|b|."""
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


# ---- deterministic synthetic PLINK data (SplitMix64 counter hash) ----------------
def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    return z ^ (z >> np.uint64(31))


def u01(seed, j, i):
    with np.errstate(over="ignore"):
        x = (np.uint64(seed) * np.uint64(0x100000001B3)) ^ (np.asarray(j, np.uint64) * np.uint64(0x9E3779B97F4A7C15))
        x = _splitmix64(x) ^ np.asarray(i, np.uint64)
        return (_splitmix64(x) >> np.uint64(11)).astype(np.float64) / float(1 << 53)


def synth_dosages(M, N, miss_rate=0.0, seed=7):
    """HWE genotypes, MAF ~ U(0.05, 0.5); returns int8 (M,N) with -3 for missing."""
    j = np.arange(M)[:, None]
    i = np.arange(N)[None, :]
    maf = 0.05 + 0.45 * u01(101 + seed, np.arange(M), 0)
    g = (u01(102 + seed, j, i) < maf[:, None]).astype(np.int8) + (u01(103 + seed, j, i) < maf[:, None]).astype(np.int8)
    if miss_rate > 0:
        g = np.where(u01(104 + seed, j, i) < miss_rate, np.int8(-3), g)
    return g


def pack_bed(g):
    """int8 dosage (count of the first allele; -3 missing) -> packed .bed rows (Geno.cpp:2838-2843)."""
    M, N = g.shape
    code = np.full(g.shape, 3, np.uint8)      # dosage 0 -> 11
    code[g == 2] = 0
    code[g == 1] = 2
    code[g == -3] = 1
    pad = (-N) % 4
    if pad:
        code = np.concatenate([code, np.zeros((M, pad), np.uint8)], axis=1)
    c = code.reshape(M, -1, 4)
    return (c[:, :, 0] | (c[:, :, 1] << 2) | (c[:, :, 2] << 4) | (c[:, :, 3] << 6)).astype(np.uint8)


def write_plink(prefix, g, chroms, P=2, ncov=2, seed=3, h2=0.2, missing_pheno=0.0, binary=False, counts=False):
    """Writes prefix.bed/.bim/.fam + prefix.pheno + prefix.covar; returns nothing."""
    M, N = g.shape
    with open(prefix + ".bed", "wb") as fh:
        fh.write(b"\x6c\x1b\x01")
        fh.write(pack_bed(g).tobytes())
    with open(prefix + ".bim", "w") as fh:
        for j in range(M):
            fh.write("%d\ts%d\t0\t%d\tA\tG\n" % (chroms[j], j, j + 1))
    with open(prefix + ".fam", "w") as fh:
        for i in range(N):
            fh.write("%d %d 0 0 0 -9\n" % (i + 1, i + 1))
    rng = np.random.default_rng(seed)
    gs = np.where(g < 0, 0, g).astype(np.float64)
    gs = (gs - gs.mean(axis=1, keepdims=True)) / (gs.std(axis=1, keepdims=True) + 1e-12)
    cov = rng.standard_normal((N, ncov))
    with open(prefix + ".covar", "w") as fh:
        fh.write("FID IID " + " ".join("C%d" % (c + 1) for c in range(ncov)) + "\n")
        for i in range(N):
            fh.write("%d %d " % (i + 1, i + 1) + " ".join("%.15g" % v for v in cov[i]) + "\n")
    ys = []
    for p in range(P):
        ncausal = min(M, 50)
        idx = rng.choice(M, ncausal, replace=False)
        beta = rng.standard_normal(ncausal) * np.sqrt(h2 / ncausal)
        y = gs[idx].T @ beta + rng.standard_normal(N) * np.sqrt(1 - h2) + 0.3 * cov[:, 0]
        ys.append(y)
    Y = np.stack(ys, axis=1)
    if binary:      # liability threshold at prevalence 0.3 -> 0/1 phenotypes
        Y = (Y > np.quantile(Y, 0.7, axis=0, keepdims=True)).astype(np.float64)
    if counts == "poisson":   # count phenotypes drawn from Poisson(exp(0.25 Y + 0.4)): regenie's own --step 1 --ct converges on these
        Y = np.random.default_rng(seed + 77).poisson(np.exp(0.25 * Y + 0.4)).astype(np.float64)
    elif counts:    # count phenotypes (--ct): Poisson-like integers with the same genetic signal (regenie's Step 1 does not converge on these)
        Y = np.floor(np.exp(0.25 * Y + 0.4))
    miss = rng.random((N, P)) < missing_pheno
    with open(prefix + ".pheno", "w") as fh:
        fh.write("FID IID " + " ".join("Y%d" % (p + 1) for p in range(P)) + "\n")
        for i in range(N):
            fh.write("%d %d " % (i + 1, i + 1) + " ".join(("NA" if miss[i, p] else "%.15g" % Y[i, p]) for p in range(P)) + "\n")


def write_t2e_pheno(path, g, ntraits=2, seed=3, missing=0.0, decimals=2):
    """Time-to-event phenotypes for the samples of write_plink(prefix, g, ...): columns T1 E1 T2 E2 ...; event times exponential with
    a hazard that carries a polygenic signal, independent exponential censoring, times rounded to `decimals` (tied event times),
    `missing` of the (time, event) pairs NA."""
    M, N = g.shape
    rng = np.random.default_rng(seed + 1000)
    gs = np.where(g < 0, 0, g).astype(np.float64)
    gs = (gs - gs.mean(axis=1, keepdims=True)) / (gs.std(axis=1, keepdims=True) + 1e-12)
    cols = []
    for p in range(ntraits):
        idx = rng.choice(M, min(M, 40), replace=False)
        lp = gs[idx].T @ (rng.standard_normal(idx.size) * np.sqrt(0.3 / idx.size))
        t_ev = rng.exponential(1.0, N) * np.exp(-lp) * (4.0 + p)
        t_c = rng.exponential(6.0 + 2 * p, N)
        tm = np.round(np.minimum(t_ev, t_c), decimals) + 10.0 ** -decimals
        ev = (t_ev <= t_c).astype(int)
        miss = rng.random(N) < missing
        cols.append((tm, ev, miss))
    with open(path, "w") as fh:
        fh.write("FID IID " + " ".join("T%d E%d" % (p + 1, p + 1) for p in range(ntraits)) + "\n")
        for i in range(N):
            fh.write("%d %d " % (i + 1, i + 1) + " ".join("NA NA" if m[i] else "%.15g %d" % (t[i], e[i]) for t, e, m in cols) + "\n")


# ---- any trait mode / CV scheme through the C ABI, and the matching oracle run ---------------------
def _use_loocv(opt, prep, force_kfold):
    return bool(opt.loocv or (opt.bt and prep.n_analyzed < 5000 and not force_kfold))   # Data.cpp:353-356


def oracle_step1_any(opt: orc.Step1Options, force_kfold: bool = False):
    """orc.run_step1 with the option of keeping K-fold CV for a small binary-trait data set (the
    reference switches BT runs below 5,000 samples to LOOCV, which would leave the BT K-fold model
    untestable at oracle-friendly sizes).  Same calls, same order as orc.run_step1."""
    bim, chrom, offs, snp_ids, prep = orc.load_inputs(opt)
    bed, _ = orc.open_bed(opt.bed + ".bed", prep.n_file)
    N, P = prep.Y.shape
    blocks = orc.chrom_blocks(chrom, bim.chr_read, opt.bsize)
    use_loocv = _use_loocv(opt, prep, force_kfold)
    h0 = orc.set_ridge_params(opt.n_ridge_l0)
    h1 = orc.set_ridge_params(opt.n_ridge_l1)
    R0 = h0.size
    lam = chrom.size * (1 - h0) / h0
    cv_sizes = None if use_loocv else orc.set_folds(prep.ind_in_analysis, opt.cv_folds)
    W = [np.zeros((N, len(blocks) * R0)) for _ in range(P)]
    for b, (c, start, bs) in enumerate(blocks):
        rows = np.asarray(bed[offs[start:start + bs]])
        G = orc.read_chunk_from_bed(rows, prep.n_file, prep.ind_ignore, prep.ind_in_analysis, opt.ref_first)
        G, _ = orc.residualize_genotypes(G, prep)
        Wb = orc.ridge_level_0_loocv(G, prep, lam) if use_loocv else orc.ridge_level_0(G, prep, cv_sizes, lam)
        for ph in range(P):
            W[ph][:, b * R0:(b + 1) * R0] = Wb[ph]
    return orc.finish_level_1(opt, prep, blocks, bim.chr_read, cv_sizes, lam, h1, W, use_loocv, [])


def gpu_step1_any(opt: orc.Step1Options, force_kfold: bool = False, inject_W=None, loco_on_device: bool = False):
    """Level 0 + level 1 through librg_step1_hip.so for QT/BT x K-fold/LOOCV.
    inject_W: optional per-phenotype N x L predictors to load instead of running level 0.
    loco_on_device: let the library assemble the LOCO rows (rg_set_loco_output) instead of the host."""
    bim, chrom, offs, snp_ids, prep = orc.load_inputs(opt)
    bed, _ = orc.open_bed(opt.bed + ".bed", prep.n_file)
    blocks = orc.chrom_blocks(chrom, bim.chr_read, opt.bsize)
    B = len(blocks)
    use_loocv = _use_loocv(opt, prep, force_kfold)
    h0 = orc.set_ridge_params(opt.n_ridge_l0)
    h1 = orc.set_ridge_params(opt.n_ridge_l1)
    lam = chrom.size * (1 - h0) / h0
    cv_sizes = None if use_loocv else orc.set_folds(prep.ind_in_analysis, opt.cv_folds)
    eng = Step1Engine(0)
    eng.set_problem(X=prep.X, Y=prep.Y, mask=prep.mask, ind_in_analysis=prep.ind_in_analysis,
                    cv_sizes=cv_sizes, lam=lam, neff=prep.Neff, n_file=prep.n_file, n_blocks_total=B,
                    max_block_size=opt.bsize, ind_ignore=prep.ind_ignore if prep.ind_ignore.any() else None,
                    ref_first=opt.ref_first)
    N, P = prep.Y.shape
    R0 = lam.size
    if inject_W is None:
        rows = [np.ascontiguousarray(bed[offs[s:s + bs]]) for (_, s, bs) in blocks]
        eng.l0_blocks_host(list(range(B)), rows)
        eng.sync()
    else:
        for b in range(B):
            for ph in range(P):
                eng.set_w(b, ph, inject_W[ph][:, b * R0:(b + 1) * R0])
    L = B * R0
    tau = np.stack([orc.tau_count(h1, L, prep.Y_raw[:, ph], prep.Neff[ph]) if opt.ct else orc.tau_from_h(h1, L, opt.bt) for ph in range(P)])
    chrcols = orc.chr_columns(blocks, bim.chr_read, R0)
    cols = [nn for (_, _, nn) in chrcols]
    conv = np.ones(P, bool)
    if loco_on_device:
        eng.set_loco_output([c for (c, _, _) in chrcols], opt.nchrom)
    if opt.bt or opt.ct:
        cs, conv, best, pred = eng.l1_bt(tau, prep.Y_raw, prep.offset, cols,
                                         niter_max_ridge=opt.niter_max_ridge,
                                         niter_max_line_search_ridge=opt.niter_max_line_search_ridge,
                                         niter_max_line_search=opt.niter_max_line_search, family=1 if opt.ct else 0)
    elif use_loocv:
        cs, best, pred = eng.l1_qt_loocv(tau, cols)
    else:
        cs, best, pred = eng.l1_qt(tau, cols)
    if loco_on_device:
        loco = [np.array(pred[ph]) for ph in range(P)]
    else:
        loco = [loco_from_predictions(pred[ph], [c for (c, _, _) in chrcols], opt.nchrom) for ph in range(P)]
    eng.close()
    return dict(cumsum=cs, best=best, pred=pred, loco=loco, converged=conv, prep=prep, tau=tau, L=L,
                use_loocv=use_loocv)


def write_synth_bgen(prefix, g, chroms, seed=1, soft=0.4):
    """prefix.bgen (layout 2, 8-bit, zlib, embedded FID_IID identifiers) + prefix.sample for the hard calls g (M,N int8, -3 missing)
    of synth_dosages, with a fraction `soft` of the calls smeared into genuine genotype probabilities (deterministic hashes), so that
    dosages are not integers and the IMPUTE info score is below 1.  Variant ids / positions / alleles as write_plink's .bim."""
    from oracle import bgen as obg
    M, N = g.shape
    j = np.arange(M)[:, None]
    i = np.arange(N)[None, :]
    p_hom = np.where(g == 2, 255, 0).astype(np.int64)          # P(two copies of the first allele), P(het) in 1/255
    p_het = np.where(g == 1, 255, 0).astype(np.int64)
    smear = u01(301 + seed, j, i) < soft
    a = (u01(302 + seed, j, i) * 90).astype(np.int64)           # mass moved away from the called genotype
    b = (u01(303 + seed, j, i) * (a + 1)).astype(np.int64)      # part of it that goes to the "next" genotype
    hom, het = p_hom.copy(), p_het.copy()
    s2, s1, s0 = smear & (g == 2), smear & (g == 1), smear & (g == 0)
    hom[s2] -= a[s2]; het[s2] += b[s2]                          # the rest goes to the other homozygote (third probability)
    het[s1] -= a[s1]; hom[s1] += b[s1]
    het[s0] += b[s0]; hom[s0] += a[s0] - b[s0]
    probs = np.stack([hom, het], axis=-1).astype(np.uint8)
    assert (hom >= 0).all() and (het >= 0).all() and (hom + het <= 255).all()
    variants = [(int(chroms[k]), k + 1, "s%d" % k, "A", "G") for k in range(M)]
    obg.write_bgen(prefix + ".bgen", probs, g < 0, variants, sample_ids=["%d_%d" % (k + 1, k + 1) for k in range(N)], compression=1)
    with open(prefix + ".sample", "w") as fh:
        fh.write("ID_1 ID_2 missing\n0 0 0\n")
        for k in range(N):
            fh.write("%d %d 0\n" % (k + 1, k + 1))


def write_synth_pgen(prefix, g, chroms, seed=1, soft=0.4):
    """prefix.pgen / .pvar / .psam for the hard calls g (M,N int8 = ALT counts, -3 missing) of synth_dosages; soft > 0 adds a dosage
    track (bit-array layout) to every variant: that fraction of the observed calls gets a 16-bit dosage near its hard call, so the
    file is read with PgenReader::Read and the MaCH r2 info score is below 1.  Variant ids / positions as write_plink's .bim,
    REF = G, ALT = A."""
    from oracle import pgen as opg
    M, N = g.shape
    codes = np.where(g < 0, 3, g).astype(np.uint8)
    dos = None
    if soft > 0:
        dos = {}
        j = np.arange(M)[:, None]
        i = np.arange(N)[None, :]
        pick = (u01(401 + seed, j, i) < soft) & (g >= 0)
        delta = (u01(402 + seed, j, i) * 5000).astype(np.int64)
        val = np.where(g == 0, delta, np.where(g == 2, 32768 - delta, 16384 + delta - 2500))
        for k in range(M):
            ids = np.flatnonzero(pick[k])
            dos[k] = (0x60, ids, val[k, ids].astype(np.uint16))
    opg.write_pgen(prefix + ".pgen", codes, np.zeros(M, np.int64), wide_vrtypes=True, dosage=dos, reclen_bytes=3, seed=seed)
    with open(prefix + ".pvar", "w") as fh:
        fh.write("#CHROM\tPOS\tID\tREF\tALT\n")
        for k in range(M):
            fh.write("%d\t%d\ts%d\tG\tA\n" % (chroms[k], k + 1, k))
    with open(prefix + ".psam", "w") as fh:
        fh.write("#FID\tIID\tSEX\n")
        for k in range(N):
            fh.write("%d\t%d\tNA\n" % (k + 1, k + 1))


def synth_rare_dosages(M, N, seed=7, lo=0.001, hi=0.01, miss_rate=0.0):
    """Hard calls with MAF ~ U(lo, hi): rare, sparse variants (the carriers-only form of regenie's approximate Firth fit needs MAC < 50)."""
    j = np.arange(M)[:, None]
    i = np.arange(N)[None, :]
    maf = lo + (hi - lo) * u01(201 + seed, np.arange(M), 0)
    g = (u01(202 + seed, j, i) < maf[:, None]).astype(np.int8) + (u01(203 + seed, j, i) < maf[:, None]).astype(np.int8)
    if miss_rate > 0:
        g = np.where(u01(204 + seed, j, i) < miss_rate, np.int8(-3), g)
    return g


def write_bed_bim(prefix, g, chroms):
    """prefix.bed / .bim only (the .fam / phenotype / covariate files of another write_plink call are reused)."""
    M, N = g.shape
    with open(prefix + ".bed", "wb") as fh:
        fh.write(b"\x6c\x1b\x01")
        fh.write(pack_bed(g).tobytes())
    with open(prefix + ".bim", "w") as fh:
        for k in range(M):
            fh.write("%d\tr%d\t0\t%d\tA\tG\n" % (chroms[k], k, k + 1))
