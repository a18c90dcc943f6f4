#!/usr/bin/env python
"""Step-1 benchmark (BASELINE.json metric: Step-1 SNPs x samples x phenos / sec).

A "step" is one complete pass of the Step-1 hot path over the synthetic workload with the packed
genotypes already resident in HBM: level 0 over every SNP block (decode/impute, FP4 matrix-core fold
Gram, fp64 multi-lambda ridge solves, out-of-fold predictions, standardisation), [N>1: all-gather of
the level-0 predictors], level 1 (fold Grams, K*R1 ridge solves, CV statistics, tau selection,
per-chromosome predictions; N>1: Gram tiles and ridge systems shared among the ranks, two all-reduces)
and the LOCO assembly on the host.

N=1 workload = BASELINE.json configs[1]: synthetic PLINK bed, 50K samples x 100K SNPs, 1 QT phenotype,
bsize 1000, 22 chromosomes (the configuration the metric is quoted on; it fits one GPU).
N>1 workload = BASELINE.json configs[2]: 500K samples x 500K SNPs, 10 QT phenotypes, bsize 1000, the SNP blocks
sharded over the N GPUs -- STRONG scaling (the total work is fixed, every rank holds 500K/N SNPs), hand-off of the level-0
predictors by ONE all-to-all by phenotype over RCCL, level 1 phenotype-sharded.  `--weak` keeps the N=1 shape per GPU
instead (100K SNPs x 50K samples x 1 phenotype per rank: all-gather + shared level 1).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--samples N] [--snps M_per_gpu] [--phenos P] [--weak] [--no-cpu]
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# hg38 autosome lengths (Mb), used only to spread SNPs over 22 chromosomes
CHR_MB = [248, 242, 198, 190, 181, 171, 159, 145, 138, 133, 135, 133, 114, 107, 102, 90, 83, 80, 59, 64, 47, 51]

PEAK = {"fp4_mfma_TOPS": 10000.0,    # MX FP4 dense (MI355X_MICROARCH.md: ~10 PF dense, ubench 9099 TF at 32x32x64)
        "i8_mfma_TOPS": 5000.0,      # 2x the bf16 dense peak (MI355X_MICROARCH.md: I8 ~2x bf16 rate; the guide's MEASURED ceiling is 3,944, this repo's ubench 4,404:
                                     # fractions quoted against 5,000 are conservative by 1.14 - 1.27x)
        "f64_mfma_TFLOPS": 78.6,     # AMD datasheet FP64 matrix (not listed in the guide)
        "bf16_mfma_TFLOPS": 2500.0,  # dense bf16 = dense fp16 (MI355X_MICROARCH.md: ~2.5 PF dense, 2495 measured)
        "hbm_GBs": 8000.0}


def snps_per_chrom(M, one_chrom=False):
    if one_chrom:
        return [M]
    w = np.array(CHR_MB, float)
    n = np.floor(M * w / w.sum()).astype(int)
    n[0] += M - n.sum()
    return n.tolist()


def gen_block(torch, dev, block_id, bs, N, seed, ncausal):
    """HWE genotypes, MAF ~ U(0.05,0.5), no missing calls; returns (packed [bs, N/4] uint8, y contribution)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed * 1000003 + block_id)
    maf = 0.05 + 0.45 * torch.rand(bs, 1, generator=g, device=dev)
    d = (torch.rand(bs, N, generator=g, device=dev) < maf).to(torch.uint8) + \
        (torch.rand(bs, N, generator=g, device=dev) < maf).to(torch.uint8)
    # bed codes (reference Geno.cpp:2838-2843): dosage 2 -> 00, 1 -> 10, 0 -> 11
    code = torch.where(d == 2, torch.zeros_like(d), torch.where(d == 1, torch.full_like(d, 2), torch.full_like(d, 3)))
    c = code.view(bs, N // 4, 4)
    packed = (c[:, :, 0] | (c[:, :, 1] << 2) | (c[:, :, 2] << 4) | (c[:, :, 3] << 6)).contiguous()
    ncausal = min(ncausal, bs)
    idx = torch.randperm(bs, generator=g, device=dev)[:ncausal]
    beta = torch.randn(ncausal, generator=g, device=dev, dtype=torch.float64)
    p = maf[idx, 0].double()
    gs = (d[idx].double() - 2 * p[:, None]) / torch.sqrt(2 * p * (1 - p))[:, None]
    return packed, beta @ gs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--samples", type=int, default=None)
    ap.add_argument("--snps", type=int, default=None, help="SNPs per GPU")
    ap.add_argument("--phenos", type=int, default=None)
    ap.add_argument("--weak", action="store_true", help="N>1: weak scaling of the N=1 workload instead of configs[2]")
    ap.add_argument("--bsize", type=int, default=1000)
    ap.add_argument("--bt", action="store_true", help="binary traits (BASELINE configs[3]'s kind): liability-threshold phenotypes, level 1 = logistic ridge (rg_l1_bt)")
    ap.add_argument("--t2e", action="store_true", help="time-to-event traits (--t2e): exponential event times whose hazard carries the polygenic signal, independent censoring; "
                    "level 1 = Cox ridge (rg_l1_cox), one call per trait; the offset of the null Cox model is taken as zero")
    ap.add_argument("--loocv", action="store_true", help="leave-one-out cross-validation at both levels (regenie --loocv: ridge_level_0_loocv, ridge_level_1_loocv / "
                    "the LOO branch of the logistic ridge) instead of 5 folds")
    ap.add_argument("--prev", default=None, help="--bt: comma-separated case prevalences of the traits (default: 5 %% ... 30 %% evenly spaced, SURVEY 8(d))")
    ap.add_argument("--oracle-trait", type=int, default=-1, help="--bt --oracle-check: the trait whose first fold chain the numpy oracle refits at full size "
                    "(default: the one with the lowest prevalence)")
    ap.add_argument("--one-chrom", action="store_true", help="all SNPs on chromosome 1 (full blocks only: per-block timings of a few blocks)")
    ap.add_argument("--l0-only", action="store_true", help="time level 0 alone (a GPU's share of a run whose W does not fit one device: 50 phenotypes at 500,000 samples)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-blocks", type=int, default=4)
    ap.add_argument("--no-ref", action="store_true", help="cpu_baseline: skip the reference binary (oracle/_ref/regenie), time the numpy oracle instead")
    ap.add_argument("--oracle-check", action="store_true", help="check the full configuration against the numpy oracle even with --no-cpu "
                    "(level-0 predictors of two blocks, level 1 of phenotype 0 on the full W: CV sums, selected ridge value, LOCO)")
    ap.add_argument("--no-disk", action="store_true", help="skip the end-to-end-from-files leg (the C++ driver on a .bed written to disk)")
    ap.add_argument("--disk-leg", action="store_true", help="run the end-to-end-from-files leg even with --no-cpu; the engine's device memory is released "
                    "before the driver starts (BASELINE configs[2] fills the device: 62.5 GB .bed, skipped with a stated reason when the disk is short)")
    ap.add_argument("--no-extra", action="store_true", help="skip the sub-records of the default N=1 run: BASELINE configs[2] in full on this one GPU "
                    "(`config3_single_gpu`) and the Step-2 record at configs[4]'s shape (`step2`)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); gloo only for the "
                    "single-box smoke of the N>1 code path")
    ap.add_argument("--single-device", action="store_true", help="test mode: every rank uses cuda:0")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from regenie_amd import hostprep as hp
    from regenie_amd.distributed import allgather_w, exchange_w_by_phenotype, shard_blocks, shard_phenotypes
    from regenie_amd.engine import Step1Engine, loco_from_predictions

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    if args.single_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    strong = world > 1 and not args.weak
    if strong:      # BASELINE configs[2]: 500K x 500K x 10 QT, total fixed
        total_snps = 500000
        args.samples = args.samples or 500000
        args.snps = args.snps or (total_snps + world - 1) // world
        args.phenos = args.phenos or 10
    else:           # BASELINE configs[1] per GPU
        args.samples = args.samples or 50000
        args.snps = args.snps or 100000
        args.phenos = args.phenos or 1
    N, P, bsize = args.samples, args.phenos, args.bsize
    assert N % 4 == 0
    M = args.snps * world
    spc = snps_per_chrom(M, args.one_chrom)
    blocks = hp.chrom_blocks(spc, bsize)
    B = len(blocks)
    R0 = R1 = 5
    shards = shard_blocks(B, world)
    b0, nb = shards[rank]
    my_blocks = list(range(b0, b0 + nb))

    # ---- synthetic data: genotypes generated in HBM, phenotype = G beta + noise ------------------
    t_gen = time.time()
    packed, ycontrib = {}, torch.zeros(P, N, dtype=torch.float64, device=dev)
    for b in my_blocks:
        pk, yc = gen_block(torch, dev, b, blocks[b][2], N, 1234, 10)
        packed[b] = pk
        ycontrib[0] += yc
        for p in range(1, P):
            ycontrib[p] += torch.roll(yc, 7919 * p)
    if world > 1:
        dist.all_reduce(ycontrib)
    rng = np.random.default_rng(99)
    cov = rng.standard_normal((N, 2))
    gval = ycontrib.cpu().numpy().T
    gval = gval / gval.std(axis=0, keepdims=True)
    h2 = 0.2
    Yraw = math_sqrt(h2) * gval + math_sqrt(1 - h2) * rng.standard_normal((N, P)) + 0.2 * cov[:, :1]
    X = hp.get_basis(np.concatenate([np.ones((N, 1)), cov], axis=1))
    mask = np.ones((N, P), bool)
    neff = np.full(P, float(N))
    bt_offset = None
    if args.bt:     # cases = liability above its (1 - prevalence) quantile, prevalence 5 - 30 %; the null logistic model on the
        prev = np.linspace(0.05, 0.3, P)          # covariates gives the offset of the level-1 logistic ridge (fit_null_logistic)
        if args.prev:
            prev = np.array([float(v) for v in args.prev.split(",")])
            assert prev.size == P, "--prev needs one value per trait"
        Yraw = np.column_stack([(Yraw[:, q] > np.quantile(Yraw[:, q], 1 - prev[q])).astype(np.float64) for q in range(P)])
        bt_offset = np.zeros((N, P))
        for q in range(P):
            b = np.zeros(X.shape[1])
            for _ in range(50):
                eta = X @ b
                pr = 1 / (1 + np.exp(-eta))
                w = pr * (1 - pr)
                step = np.linalg.solve(X.T @ (X * w[:, None]), X.T @ (Yraw[:, q] - pr))
                b += step
                if np.abs(step).max() < 1e-10:
                    break
            bt_offset[:, q] = X @ b
    t2e_event = None
    if args.t2e:    # time = min(event time, censoring time); the time column is also the level-0 response (Pheno.cpp:262-283, Step1_Models.cpp:2259)
        t_ev = rng.exponential(1.0, (N, P)) * np.exp(-0.5 * (Yraw - Yraw.mean(axis=0)) / Yraw.std(axis=0)) * 4.0
        t_c = rng.exponential(6.0, (N, P))
        t2e_event = (t_ev <= t_c).astype(np.float64)
        Yraw = np.minimum(t_ev, t_c)
    Y, _ = hp.residualize_pheno(Yraw - Yraw.mean(axis=0), X, mask, neff)
    ain = np.ones(N, bool)
    cv_sizes = None if args.loocv else hp.set_folds(ain, 5)
    lam = M * (1 - hp.set_ridge_params(R0)) / hp.set_ridge_params(R0)
    L = B * R0
    h1 = hp.set_ridge_params(R1)
    tau = np.tile(L * (1 - h1) / h1 * (3.0 / np.pi ** 2 if args.bt else 1.0), (P, 1))     # check_l0 (Step1_Models.cpp:2115-2117)
    cols_per_chr = [sum(1 for bl in blocks if bl[0] == c) * R0 for c in range(len(spc))]
    chroms = [c + 1 for c, n in enumerate(cols_per_chr) if n > 0]
    cols_per_chr = [n for n in cols_per_chr if n > 0]
    t_gen = time.time() - t_gen

    pheno_sharded = world > 1 and P >= world
    eng = Step1Engine(local, torch.cuda.current_stream().cuda_stream)
    eng.set_problem(X=X, Y=Y, mask=mask, ind_in_analysis=ain, cv_sizes=cv_sizes, lam=lam, neff=neff,
                    n_file=N, n_blocks_total=B, max_block_size=bsize)
    if pheno_sharded:           # this rank only ever produces (and sends on) the predictors of its own blocks
        eng.set_block_range(b0, nb)
    Wt = torch.zeros(eng.w_bytes // 8, dtype=torch.float64, device=dev)
    eng.set_w_buffer(Wt.data_ptr(), eng.w_bytes)
    Wv = Wt.view(nb * R0 if pheno_sharded else L, P, eng.w_rows)      # phenotype-sharded: rows of this rank's blocks only
    ptrs = [packed[b].data_ptr() for b in my_blocks]
    bss = [blocks[b][2] for b in my_blocks]
    eng.set_loco_output(chroms)                             # level 1 returns the 23 LOCO rows (write_predictions' assembly)

    class _DevBuf:                                          # raw device pointer -> torch tensor view
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 3}

    def _allreduce(ptr, n):                                 # completes the level-1 Gram / solution buffers
        t = torch.as_tensor(_DevBuf(ptr, n), device=dev)
        if args.backend == "nccl":
            dist.all_reduce(t)
        else:                                               # gloo (CPU test mode): stage through host memory
            h = t.cpu()
            dist.all_reduce(h)
            t.copy_(h)
        torch.cuda.synchronize()

    # N>1 hand-off of the level-0 predictors: with at least one phenotype per rank the ranks exchange predictor slabs
    # BY PHENOTYPE (all-to-all, 1/world of the all-gather volume) and each runs level 1 for its own phenotypes; with
    # fewer phenotypes than ranks W is all-gathered and level 1 is shared tile-wise (two all-reduces).
    pshards = shard_phenotypes(P, world) if pheno_sharded else None
    xbuf = {}                                               # exchange buffers, allocated once
    if world > 1 and not pheno_sharded:
        eng.set_collective(world, rank, _allreduce)

    extra_t = {}

    def step(exchange=True, solo=False):
        eng.l0_blocks_device(my_blocks, bss, ptrs, N // 4)
        eng.sync()
        if pheno_sharded and not solo:
            Wg = exchange_w_by_phenotype(Wv, shards, pshards, R0, via_host=(args.backend != "nccl"), buffers=xbuf, own_rows_only=True)
            torch.cuda.synchronize()
            q0, qn = pshards[rank]
            eng.set_l1_view(Wg.data_ptr(), q0, qn)
            cs, best, pred = eng.l1_qt(tau[q0:q0 + qn], cols_per_chr)
            eng.set_l1_view(None, 0, P)
            mine = ([pred[p] for p in range(qn)], cs, [int(b) for b in best])          # LOCO rows, assembled on the device
            gathered = [None] * world                        # small per-phenotype summaries; predictions stay on their rank
            dist.all_gather_object(gathered, (float(sum(np.abs(l).sum() for l in mine[0])), mine[2]))
            return (mine[0], cs, [b for g in gathered for b in g[1]], sum(g[0] for g in gathered))
        if world > 1 and exchange and not pheno_sharded:
            allgather_w(Wv, shards, R0, force_broadcast=(args.backend != "nccl"))
            torch.cuda.synchronize()
        out = None
        if solo:                                            # rank-0-only timing pass: no collective may be issued
            eng.set_collective(1, 0, None)
            if pheno_sharded:
                return None                                 # W is not gathered in this mode: level 0 only
        if args.l0_only:
            return ([np.zeros((N, 1))], None, [0] * P)
        if rank == 0 or (world > 1 and not solo):           # N>1: level 1 is shared among the ranks
            t_l1 = time.perf_counter()
            if args.bt:
                if args.oracle_check and not args.loocv:     # the fold models' coefficients come back too (rg_bt_options.beta_out): what the oracle leg compares
                    cs, conv, best, pred, fold_betas, fold_cs = eng.l1_bt(tau, Yraw, bt_offset, cols_per_chr, fold_detail=True)
                    extra_t["_fold_detail"] = (fold_betas, fold_cs)
                else:
                    cs, conv, best, pred = eng.l1_bt(tau, Yraw, bt_offset, cols_per_chr)
                extra_t["bt_converged"] = [bool(c) for c in conv]
            elif args.t2e:
                pred, best, cs, conv, taus = [], [], [], [], []
                for q in range(P):
                    tq, dq, cq, bq, pq = eng.l1_cox(q, Yraw[:, q], t2e_event[:, q], np.zeros(N), cols_per_chr)
                    pred.append(pq); best.append(bq); cs.append(dq); conv.append(cq); taus.append(tq)
                extra_t["t2e"] = {"converged": [bool(c) for c in conv], "tau": [list(map(float, t)) for t in taus],
                                  "held_out_deviance": [list(map(float, d)) for d in cs], "best": [int(b) for b in best],
                                  "events_fraction": float(t2e_event.mean())}
            elif args.loocv:
                cs, best, pred = eng.l1_qt_loocv(tau, cols_per_chr)
            else:
                cs, best, pred = eng.l1_qt(tau, cols_per_chr)
            extra_t["level1_wall_ms_last_step"] = (time.perf_counter() - t_l1) * 1e3
            out = [pred[p] for p in range(P)], cs, best                                    # LOCO rows, assembled on the device
        return out

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    import threading

    class _MemPeak(threading.Thread):                       # device memory in use, sampled while the steps run (level-1 buffers live only inside the calls)
        def __init__(self):
            super().__init__(daemon=True)
            self.peak, self.stop = 0, False

        def run(self):
            while not self.stop:
                fr, tot = torch.cuda.mem_get_info(dev)
                self.peak = max(self.peak, tot - fr)
                time.sleep(0.01)
    mem = _MemPeak()
    mem.start()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    sec_per_step = dt / args.steps
    value = M * N * P / sec_per_step
    mem.stop = True
    mem.join()
    extra_t["device_memory_peak_GB"] = mem.peak / 1e9      # includes the resident synthetic genotypes and W

    # ---- per-kernel timing pass (HIP events on the ctx stream) + roofline of the dominant kernel ----
    roof, kernels = None, None
    if rank == 0:
        eng.enable_timing(True)
        step(exchange=False, solo=True)   # rank-0-only pass: no collective may be issued here (W is already gathered)
        tm = eng.timing()
        eng.enable_timing(False)
        n_batches = max(1, int(tm["n_gram_launches"]))      # one Gram launch per level-0 batch
        bs_eff = float(np.mean(bss))
        flops = {
            "gram_fp4": 2.0 * N * sum(x * x for x in bss),                       # F_gram = 2 N bs^2 per block
            # K*R0 systems per block; leave-one-out: R0 systems whose forward substitution carries the N sample rows (loocv.hip)
            # leave-one-out (loocv_tri.hip): ONE Householder tridiagonal reduction per block (4/3 bs^3) + the GEMM Z = Q^T G~ (2 N bs^2) serve every
            # ridge value; the recurrences (N bs R0 (3 + P) multiply-adds) are vector work and not counted here
            "chol_f64": (sum(4.0 * x ** 3 / 3.0 + 2.0 * N * x * x for x in bss) if args.loocv else
                         sum((x ** 3 / 3.0 + 2.0 * x * x * P) * 5 * R0 for x in bss)),
            # level-1 fold Grams: the symmetric product, lower triangle with the diagonal (what any algorithm must form); SURVEY 8(d)'s
            # 2 N L^2 P counts the full product the reference's W_i^T W_i computes -- twice this, reported next to it
            "l1_gram_f64": 1.0 * N * L * (L + 1) * P,
            "l1_chol_f64": P * (1 if args.loocv else 5) * R1 * (L ** 3 / 3.0 + 2.0 * L * L),
            # many-row predictions on the i8 matrix cores: 8 digit planes x 2 N bs (P R0) integer operations per block
            "pred_i8": 8 * 2.0 * N * sum(bss) * P * R0 if P * R0 > 16 else 0.0,
            # the iterative level-1 models (--bt / --t2e): one weighted Gram X^T W X per fold model and IRLS step, the symmetric
            # product over the positions it contracts, counted as EXECUTED (rg_timing.wgram_positions sums them over the Grams)
            "wgram_f64": 1.0 * tm["wgram_positions"] * L * (L + 1),
        }
        kernels = {
            "gram_fp4": {"ms": tm["ms_gram"], "achieved_TOPS": flops["gram_fp4"] / (tm["ms_gram"] * 1e-3) / 1e12 if tm["ms_gram"] else None,
                        "launches": tm["n_gram_launches"]},
            "chol_f64": {"ms": tm["ms_chol"], "achieved_TFLOPS": flops["chol_f64"] / (tm["ms_chol"] * 1e-3) / 1e12 if tm["ms_chol"] else None},
            "prep": {"ms": tm["ms_prep"]}, "geno_xy": {"ms": tm["ms_xy"]}, "assemble_form": {"ms": tm["ms_assemble"]},
            "pred": {"ms": tm["ms_pred"], "achieved_TOPS_i8_digit_planes": flops["pred_i8"] / (tm["ms_pred"] * 1e-3) / 1e12 if (tm["ms_pred"] and flops["pred_i8"]) else None},
            "l1_gram_f64": {"ms": tm["ms_l1_gram"], "achieved_TFLOPS": flops["l1_gram_f64"] / (tm["ms_l1_gram"] * 1e-3) / 1e12 if tm["ms_l1_gram"] else None,
                            "full_product_TFLOPS_survey_8d": 2.0 * N * L * L * P / (tm["ms_l1_gram"] * 1e-3) / 1e12 if tm["ms_l1_gram"] else None},
            "l1_chol_f64": {"ms": tm["ms_l1_chol"]}, "l1_cv_pred": {"ms": tm["ms_l1_pred"]},
        }
        wg_bf16 = tm["n_wgram_approx_rounds"] > 0      # the quasi-Newton Grams of wgram_bf16.hip
        wg_mult = 3.0 if os.environ.get("RG_WGRAM_FMT") == "bf16x3" else 1.0      # products executed per operand pair: fp16 plane (default) or bf16 hi + lo
        if tm["n_wgram"]:
            kernels["wgram_f64"] = {"ms": tm["ms_wgram"], "achieved_TFLOPS": flops["wgram_f64"] / (tm["ms_wgram"] * 1e-3) / 1e12 if tm["ms_wgram"] else None,
                                    "chain_grams": tm["n_wgram"], "irls_rounds": tm["n_irls_rounds"], "grams_per_trait": tm["n_wgram"] / P,
                                    "ms_per_chain_gram": tm["ms_wgram"] / tm["n_wgram"],
                                    "quasi_newton_bf16_rounds": tm["n_wgram_approx_rounds"],
                                    "note": ("achieved_TFLOPS counts the symmetric product positions x L x (L + 1) per chain Gram; the kernel executes "
                                             + ("3x that in bf16 (hi hi^T + hi lo^T + lo hi^T)" if wg_mult == 3.0 else "exactly that, once, on fp16 operands") +
                                             "; conversion and slice reduction included in ms") if wg_bf16 else
                                            "fp64 matrix cores (k_wgram128), slice reduction and the X^T W z row included in ms"}
            kernels["irls_solve"] = {"ms": tm["ms_irls_solve"]}
            # the streaming half of an IRLS round: eta = X beta over the samples (k_bt_eval) and the exact score X^T (y - p) - tau beta
            # (k_bt_score), one pass over the phenotype's L x N predictors each (8 L N bytes; the vectors they carry are 1e-3 of that)
            stream_bytes = 8.0 * L * N * tm.get("n_irls_passes", 0)
            kernels["irls_stream"] = {"ms": tm["ms_irls_stream"], "passes": tm.get("n_irls_passes", 0),
                                      "achieved_GBps": stream_bytes / (tm["ms_irls_stream"] * 1e-3) / 1e9 if tm["ms_irls_stream"] else None}
        cand = (["gram_fp4", "chol_f64", "l1_gram_f64"] + (["pred"] if flops["pred_i8"] else []) + (["wgram_f64"] if tm["n_wgram"] and tm["ms_wgram"] else []) +
                (["irls_stream"] if tm["n_wgram"] and tm.get("n_irls_passes", 0) else []))
        dom = max(cand, key=lambda k: kernels[k]["ms"])
        kernels["gram_fp4"]["frac_of_fp4_peak"] = kernels["gram_fp4"]["achieved_TOPS"] / PEAK["fp4_mfma_TOPS"]
        traffic, traffic_note = measured_traffic(dom, len(my_blocks), n_batches, P)
        if dom == "gram_fp4":
            a = kernels[dom]["achieved_TOPS"]
            roof = {"kernel": "k_gram_fp4_blocks (FP4 matrix-core fold Gram)", "bound": "mfma", "achieved": a, "peak": PEAK["fp4_mfma_TOPS"],
                    "unit": "TOP/s", "frac": a / PEAK["fp4_mfma_TOPS"], "traffic": traffic, "traffic_note": traffic_note,
                    "algorithmic_ops_per_launch": flops["gram_fp4"] / max(1, n_batches), "avg_launch_ms": tm["ms_gram"] / max(1, n_batches)}
        elif dom == "pred":
            a = kernels[dom]["achieved_TOPS_i8_digit_planes"]
            roof = {"kernel": "k_l0_pred_i8 (level-0 predictions, exact digit planes on the i8 matrix cores)", "bound": "mfma", "achieved": a,
                    "peak": PEAK["i8_mfma_TOPS"], "unit": "TOP/s", "frac": a / PEAK["i8_mfma_TOPS"], "traffic": traffic, "traffic_note": traffic_note,
                    "algorithmic_ops_per_launch": flops["pred_i8"] / max(1, n_batches), "avg_launch_ms": tm["ms_pred"] / max(1, n_batches)}
        elif dom == "irls_stream":
            a = kernels[dom]["achieved_GBps"]
            roof = {"kernel": "k_bt_eval + k_bt_score (the streaming passes of the logistic-ridge IRLS rounds over the phenotype's predictors; the Hessians are "
                              "the quasi-Newton Grams of wgram_bf16.hip, reused over several rounds)", "bound": "hbm", "achieved": a, "peak": PEAK["hbm_GBs"],
                    "unit": "GB/s", "frac": a / PEAK["hbm_GBs"], "traffic": traffic, "traffic_note": traffic_note,
                    "algorithmic_bytes_per_launch": 8.0 * L * N, "avg_launch_ms": tm["ms_irls_stream"] / max(1, tm["n_irls_passes"]),
                    "note": "avg_launch_ms includes the small host <-> device copies and the synchronisation each pass ends with"}
        elif dom == "wgram_f64" and wg_bf16:
            ex = wg_mult * flops["wgram_f64"] / (tm["ms_wgram"] * 1e-3) / 1e12     # 16-bit matrix operations executed per second
            roof = {"kernel": "k_wsplit + k_wgram_mx + k_wg_reduce (quasi-Newton weighted Gram of the logistic ridge IRLS: " +
                              ("bf16 pair planes, three products per operand pair" if wg_mult == 3.0 else "one fp16 operand plane, one product per operand pair") +
                              ", fp32 accumulators flushed into fp64 partial tiles; the score that drives the iteration and its stopping rule stays exact fp64)",
                    "bound": "mfma", "achieved": ex, "peak": PEAK["bf16_mfma_TFLOPS"],
                    "unit": "TFLOP/s (%s executed)" % ("bf16" if wg_mult == 3.0 else "fp16"), "frac": ex / PEAK["bf16_mfma_TFLOPS"], "traffic": traffic, "traffic_note": traffic_note,
                    "fp64_equivalent_TFLOPS": kernels[dom]["achieved_TFLOPS"],
                    # a launch = one round in which Grams were formed (k_wsplit + k_wgram_mx + k_wg_reduce over the chains that needed one);
                    # the other IRLS rounds step on stored Hessians and form none
                    "algorithmic_flops_per_launch": wg_mult * flops[dom] / max(1, tm["n_wgram_approx_rounds"]),
                    "avg_launch_ms": kernels[dom]["ms"] / max(1, tm["n_wgram_approx_rounds"])}
        else:
            a = kernels[dom]["achieved_TFLOPS"]
            nlaunch = {"chol_f64": n_batches, "l1_gram_f64": P, "wgram_f64": tm["n_irls_rounds"]}[dom]
            roof = {"kernel": {"chol_f64": "k_c128_panel x (order / 128) + k_chol_backsolve (fp64 MFMA batched Cholesky by panels of 128 columns, chol_p128.h, per level-0 batch of systems)",
                               "l1_gram_f64": "k_l1_gram128 (fp64 MFMA level-1 fold Gram, one launch per phenotype)",
                               "wgram_f64": "k_wgram128 (fp64 MFMA weighted Gram X^T W X of the logistic / Cox ridge IRLS, one launch per lock-step round "
                                            "over the unfinished fold models)"}[dom], "bound": "mfma", "achieved": a,
                    "peak": PEAK["f64_mfma_TFLOPS"], "unit": "TFLOP/s", "frac": a / PEAK["f64_mfma_TFLOPS"], "traffic": traffic,
                    "traffic_note": traffic_note,
                    "algorithmic_flops_per_launch": flops[dom] / max(1, nlaunch),
                    "avg_launch_ms": kernels[dom]["ms"] / max(1, nlaunch)}

    # ---- CPU baseline: the oracle (numpy/OpenBLAS restatement of the reference) on a bounded sample ----
    cpu = None
    bt_check = None
    if rank == 0 and world == 1 and args.bt and args.oracle_check and not args.loocv:
        # the logistic ridge of ONE fold chain refitted by the numpy oracle at this run's full size.  The device is handed back first (the
        # default line runs this sub-run next to its other sub-records: RG_GPU_DONE on stderr tells the parent the GPU is free)
        q = args.oracle_trait if args.oracle_trait >= 0 else int(np.argmin(Yraw.mean(axis=0)))
        W0 = np.concatenate([eng.get_w(b, q) for b in range(B)], axis=1)
        fold_betas, fold_cs = extra_t.pop("_fold_detail")
        eng.close()
        eng = None
        packed.clear()
        Wt = Wv = None
        torch.cuda.empty_cache()
        print("RG_GPU_DONE", file=sys.stderr, flush=True)
        bt_check = bt_oracle_leg(W0, Yraw[:, q], bt_offset[:, q], mask[:, q], cv_sizes, tau[q], q, fold_betas[q, 0], fold_cs[q, 0],
                                 float(Yraw[:, q].mean()), bool(kernels.get("wgram_f64", {}).get("quasi_newton_bf16_rounds")))
        del W0
    elif rank == 0 and world == 1 and args.bt:
        extra_t.pop("_fold_detail", None)
    elif rank == 0 and world == 1 and args.loocv and args.oracle_check:
        # the leave-one-out level 0 at THIS run's block order and sample count against the oracle's eigendecomposition route (checker only)
        from oracle import regenie_step1 as orc
        prep = orc.Prepared(ids=[], n_file=N, ind_ignore=np.zeros(N, bool), ind_in_analysis=ain, pheno_names=[], Y=Y, Y_raw=None, mask=mask, X=X,
                            Neff=neff, scale_Y=np.ones(P), ncov=X.shape[1], n_analyzed=N)
        osel = [my_blocks[0], my_blocks[-1]] if len(my_blocks) > 1 else [my_blocks[0]]
        werr, t_or = 0.0, 0.0
        for b in osel:
            rows = packed[b].cpu().numpy()
            t0 = time.perf_counter()
            G = orc.read_chunk_from_bed(rows, N, None, ain)
            G, _ = orc.residualize_genotypes(G, prep)
            Wb = orc.ridge_level_0_loocv(G, prep, lam)
            t_or += time.perf_counter() - t0
            del G
            for p in range(P):
                werr = max(werr, float(np.max(np.abs(eng.get_w(b, p) - Wb[p])) / np.max(np.abs(Wb[p]))))
            del Wb
        extra_t["level0_vs_oracle"] = {"W_max_rel_err": werr, "blocks": len(osel), "block_order": [blocks[b][2] for b in osel], "samples": N, "phenos": P,
                                       "oracle": "oracle.regenie_step1.ridge_level_0_loocv (eigendecomposition route, Step1_Models.cpp:615-726)", "oracle_s": t_or}
    elif rank == 0 and world == 1 and (not args.no_cpu or args.oracle_check) and not args.loocv and not args.t2e:
        cpu = cpu_baseline(args, eng, torch, dev, packed, blocks, my_blocks, X, Y, Yraw, cov, mask, ain, neff, cv_sizes, tau, M, N, P, B, R0,
                           (res[0], res[1], res[2]), with_reference=not (args.no_cpu or args.no_ref))

    disk = None
    if rank == 0 and world == 1 and not args.no_cpu and not args.no_disk:
        disk = from_disk_leg(args, torch, packed, blocks, my_blocks, Yraw, cov, M, N, P, res[0])
    elif rank == 0 and world == 1 and args.disk_leg:      # forced (the configs[2] sub-run of the default line): the device is handed over first
        loco_ck_forced = float(sum(np.abs(l).sum() for l in res[0]))
        best_forced = [int(b) for b in res[2]]

        def _free():
            nonlocal eng, Wt, Wv
            eng.close()
            eng = None
            packed.clear()
            Wt = Wv = None
            torch.cuda.empty_cache()
        disk = from_disk_leg(args, torch, packed, blocks, my_blocks, Yraw, cov, M, N, P, res[0], free_gpu=_free, nruns=2)

    # ---- sub-records of the default N=1 run (the engine's memory is released first) ----
    extra = {}
    p4 = None
    default_n1 = rank == 0 and world == 1 and not args.no_cpu and not args.no_extra and (N, M, P) == (50000, 100000, 1)
    if default_n1:
        loco_ck = float(sum(np.abs(l).sum() for l in res[0]))
        eng.close()
        eng = None
        del packed, Wt, Wv
        torch.cuda.empty_cache()
        import subprocess
        me = [sys.executable, os.path.abspath(__file__)]
        big = ["--samples", "500000", "--steps", "1", "--no-cpu"]

        def sub_line(argv, timeout):
            r = subprocess.run(me + argv, capture_output=True, text=True, timeout=timeout)
            js = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if not js:
                raise RuntimeError((r.stdout + r.stderr)[-600:])
            return json.loads(js[-1])
        # BASELINE configs[3]'s kind (binary traits) at its level-1 shape: 500,000 samples, L = 2,560 level-0 predictors (512 blocks of 100 SNPs --
        # level 1 does not see the block width), four traits of prevalence 5 %, 30 %, 1 % and 50 %; 98 % of that configuration is this level 1
        # (DESIGN.md section 5).  With --oracle-check the sub-run then hands the device back (RG_GPU_DONE on stderr) and has the numpy oracle refit
        # the first fold chain of the 1 % trait at full size on the host -- minutes of host work that run NEXT TO the GPU-bound sub-records
        # that follow (the leave-one-out level 0, the Step-2 kernels); configs[2] and the leave-one-out level 1 above and the BGEN run below
        # (all host-sensitive: the job has 16 CPUs of quota) have the box to themselves.
        try:    # BASELINE configs[2] (the north star's target workload) in full on this ONE GPU: 500,000 x 500,000 x 10 QT, resident
            l3 = sub_line(big + ["--snps", "500000", "--phenos", "10", "--warmup", "1"] + ([] if args.no_disk else ["--disk-leg"]), 1500)
            extra["config3_single_gpu"] = {k: l3[k] for k in ("ms_per_step", "value", "unit", "steps", "warmup", "loco_checksum", "selected_tau_index", "roofline",
                                                               "kernels", "end_to_end_from_files", "config")}
        except Exception as e:   # noqa: BLE001
            extra["config3_single_gpu"] = {"error": repr(e)[:500]}
        # leave-one-out cross-validation (regenie --loocv) at the target sample count, level 1 at L = 2,560 for two quantitative traits and for one
        # binary trait: the level-1 solvers wait on the host between steps, so they too run before the oracle takes the CPUs
        lo = {}
        try:
            lq = sub_line(big + ["--loocv", "--snps", "51200", "--bsize", "100", "--phenos", "2", "--warmup", "0"], 900)
            lo["level1_qt_s_per_trait"] = lq["level1"]["level1_wall_ms_last_step"] / 2e3
            lo["level1_qt_device_memory_peak_GB"] = lq["level1"].get("device_memory_peak_GB")
            lo["level1_qt_selected_tau_index"] = lq["selected_tau_index"]
            lb = sub_line(big + ["--loocv", "--snps", "51200", "--bsize", "100", "--phenos", "1", "--bt", "--prev", "0.1", "--warmup", "0"], 900)
            lo["level1_bt_s_per_trait"] = lb["level1"]["level1_wall_ms_last_step"] / 1e3
            lo["level1_bt_converged"] = lb["level1"].get("bt_converged")
            lo["level1_bt_device_memory_peak_GB"] = lb["level1"].get("device_memory_peak_GB")
        except Exception as e:   # noqa: BLE001
            lo["error_level1"] = repr(e)[:500]
        # Step-2 kernels (device time between the library's events -- but the missing-call route waits for a 4-byte count on the host between
        # two kernels: beside the oracle refit below, whose numpy threads exhaust the job's 16-CPU quota, that wait grew from 0.02 to 12 ms
        # per block in round 6's first line): measured while the box is quiet
        try:
            from tools.step2_record import step2_record
            extra["step2"] = step2_record(torch=torch)
        except Exception as e:   # noqa: BLE001
            extra["step2"] = {"error": repr(e)[:500]}
        p4, err4 = None, []
        try:
            p4 = subprocess.Popen(me + big + ["--snps", "51200", "--bsize", "100", "--phenos", "4", "--bt", "--prev", "0.05,0.3,0.01,0.5", "--warmup", "0",
                                              "--oracle-check", "--oracle-trait", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            gpu_free = threading.Event()

            def _drain():
                for ln in p4.stderr:
                    err4.append(ln)
                    if "RG_GPU_DONE" in ln:
                        gpu_free.set()
                gpu_free.set()
            threading.Thread(target=_drain, daemon=True).start()
            gpu_free.wait(timeout=900)
        except Exception as e:   # noqa: BLE001 - a sub-record must not take the line down
            extra["config4_level1_binary_traits"] = {"error": repr(e)[:500]}
        # leave-one-out level 0 on eight full blocks of 1,000 SNPs with ten traits (device time: it does not mind the oracle beside it)
        try:
            l0 = sub_line(big + ["--loocv", "--snps", "8000", "--one-chrom", "--phenos", "10", "--l0-only", "--warmup", "1", "--oracle-check"], 900)
            lo["level0_ms_per_block_of_1000_snps"] = l0["ms_per_step"] / 8
            lo["level0_device_memory_peak_GB"] = l0["level1"].get("device_memory_peak_GB")
            lo["level0_kernels"] = {k: l0["kernels"][k] for k in ("prep", "gram_fp4", "assemble_form", "chol_f64", "pred")}
            lo["level0_vs_oracle"] = l0["level1"].get("level0_vs_oracle")
            lo["config"] = "500,000 samples; level 0: 8 blocks x 1,000 SNPs x 10 QT; level 1: 512 blocks x 5 ridge values (bsize 100), 2 QT / 1 BT (prevalence 10 %)"
        except Exception as e:   # noqa: BLE001
            lo["error"] = repr(e)[:500]
        extra["loocv_500k"] = lo
        if p4 is not None and "config4_level1_binary_traits" not in extra:
            try:
                out4, _ = p4.communicate(timeout=1500)
                js = [ln for ln in out4.splitlines() if ln.startswith("{")]
                if not js:
                    raise RuntimeError(("".join(err4))[-600:])
                l4 = json.loads(js[-1])
                ms4 = l4["level1"].get("level1_wall_ms_last_step", 0.0)
                extra["config4_level1_binary_traits"] = {
                    "s_per_trait": ms4 / 4e3, "level1_wall_ms": ms4, "traits": 4, "prevalences": [0.05, 0.3, 0.01, 0.5],
                    "converged": l4["level1"].get("bt_converged"), "oracle_check": l4.get("bt_oracle_check"),
                    "selected_tau_index": l4["selected_tau_index"], "loco_checksum": l4["loco_checksum"],
                    "roofline": l4["roofline"], "kernels": {k: l4["kernels"].get(k) for k in ("wgram_f64", "irls_solve", "irls_stream")}, "config": l4["config"]}
            except Exception as e:   # noqa: BLE001
                try:
                    p4.kill()
                except Exception:   # noqa: BLE001
                    pass
                extra["config4_level1_binary_traits"] = {"error": repr(e)[:500]}
        if not args.no_disk:
            # Step 2 from configs[4]'s real input format: a BGEN v1.2 file at 500,000 samples written to /tmp, `regenie-amd --step 2 --bgen` from
            # process start to exit with the shares of its block loop, regenie itself (oracle/_ref) on a bounded sample of the same encoding
            try:
                import tempfile
                with tempfile.TemporaryDirectory() as td:
                    # 18,432 variants on two chromosomes: whole batches of the device decoder (3,072 streams), as the 450,000 variants of a
                    # chromosome of BASELINE configs[4] give; 1,024 variants for regenie itself
                    rb = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "bgen_e2e.py"), "--json",
                                         os.path.join(td, "rec.json"), "500000", "18432", "1024"], capture_output=True, text=True, timeout=900,
                                        env=dict(os.environ, BGEN_E2E_NCHR="2"))
                    if rb.returncode != 0:
                        raise RuntimeError((rb.stdout + rb.stderr)[-600:])
                    extra["step2"]["bgen_from_file"] = json.load(open(os.path.join(td, "rec.json")))
                    br = extra["step2"]["bgen_from_file"].pop("bed_reference", None)
                    if br:      # regenie itself on a bounded .bed sample at the record's sample count: the record's CPU baseline; the numpy port's stays beside it
                        extra["step2"]["cpu_baseline_port"] = extra["step2"].get("cpu_baseline")
                        extra["step2"]["cpu_baseline"] = {"value": br["value"], "unit": br["unit"], "cores": br["threads"], "kind": "reference", "sample": br["sample"],
                                                          "variants_per_s": br["variants_per_s"], "result_lines_identical": "%d/%d" % (br["byte_identical"], br["result_lines"])}
            except Exception as e:   # noqa: BLE001
                extra.setdefault("step2", {})["bgen_from_file"] = {"error": repr(e)[:600]}
    if rank == 0:
        line = {
            "metric": "Step-1 SNPs x samples x phenos / sec", "value": value, "unit": "SNP*sample*pheno/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": sec_per_step * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "fp4 (exact integer Gram) + i8 (exact fixed-point digit planes of the fp64 operands: G~X / G~Y, many-row predictions) + f64 (solves, level 1)",
            "data": "synthetic",
            "config": {"workload": "%ssynthetic PLINK bed %d samples x %d SNPs (%d per GPU), %d %s pheno, bsize %d, 22 chromosomes, 5-fold CV, 5x5 ridge grid; "
                       "`value` is the HBM-resident figure (packed genotypes generated on the device, timed region = level 0 + level 1 + LOCO rows); "
                       "the complete run from files on disk is `end_to_end_from_files`"
                       % ("BASELINE configs[2], blocks sharded over the GPUs: " if strong else
                          ("BASELINE configs[1]: " if (world == 1 and (N, M, P) == (50000, 100000, 1)) else
                           ("BASELINE configs[2] on ONE GPU: " if (world == 1 and (N, M, P) == (500000, 500000, 10)) else
                            ("" if world == 1 else "weak scaling of BASELINE configs[1]: "))),
                          N, M, args.snps, P, "BT" if args.bt else ("time-to-event" if args.t2e else "QT"), bsize), "samples": N, "snps": M, "phenos": P, "bsize": bsize, "blocks": B,
                       "parallelism": ("blocks sharded x%d, all-to-all of W by phenotype, level 1 phenotype-sharded x%d" if pheno_sharded else
                                       "blocks sharded x%d, all-gather of W, level-1 Gram tiles / ridge systems shared x%d") % (world, world)},
            "roofline": roof, "kernels": kernels, "cpu_baseline": cpu, "bt_oracle_check": bt_check,
            "loco_max_rel_err": cpu.get("loco_max_rel_err") if cpu else None,
            "end_to_end_from_files": disk,
            "loco_checksum": float(res[3]) if len(res) > 3 else (loco_ck if default_n1 else float(sum(np.abs(l).sum() for l in res[0]))),
            "selected_tau_index": [int(b) for b in res[2]],
            "setup_s": {"generate": t_gen}, "level1": extra_t,
        }
        line.update(extra)
        if extra:
            line["summary"] = summary_of(line)      # last key of the line: a reader that keeps only the line's tail still gets every sub-run's headline
        print(json.dumps(line))
    if eng is not None:
        eng.close()
    if world > 1:
        dist.destroy_process_group()


def summary_of(line):
    """Compact digest (< 1,500 characters) of the sub-records, placed LAST in the JSON line."""
    def g(d, *ks):
        for k in ks:
            if not isinstance(d, dict) or d.get(k) is None:
                return None
            d = d[k]
        return round(d, 4 - int(math.floor(math.log10(abs(d)))) - 1) if isinstance(d, float) and d != 0 and math.isfinite(d) else d
    c3, c4, lo, s2 = (line.get(k) or {} for k in ("config3_single_gpu", "config4_level1_binary_traits", "loocv_500k", "step2"))
    bg = (s2.get("bgen_from_file") or {})
    run0 = (bg.get("runs") or [{}])[0] if isinstance(bg.get("runs"), list) else {}
    cases = s2.get("cases") or {}
    cb0 = (bg.get("cpu_baseline") or [{}])[0] if isinstance(bg.get("cpu_baseline"), list) else {}
    out = {
        "cfg1": {"ms": g(line, "ms_per_step"), "chol_frac": g(line, "roofline", "frac"), "traffic_B": g(line, "roofline", "traffic"),
                 "files_s": g(line, "end_to_end_from_files", "wall_s"), "loco_err": g(line, "loco_max_rel_err")},
        "cfg2_1gpu": {"ms": g(c3, "ms_per_step"), "value": g(c3, "value"), "l1_gram_frac": g(c3, "roofline", "frac"), "chol_TF": g(c3, "kernels", "chol_f64", "achieved_TFLOPS"),
                      "pred_ms": g(c3, "kernels", "pred", "ms"), "files_s": g(c3, "end_to_end_from_files", "wall_s"), "loco_ck": c3.get("loco_checksum")},
        "cfg3_bt": {"s_per_trait": g(c4, "s_per_trait"), "beta_err": g(c4, "oracle_check", "beta_max_rel_err"), "pred_err": g(c4, "oracle_check", "prediction_max_rel_err"),
                    "stream_frac": g(c4, "roofline", "frac")},
        "loocv": {"l0_ms_per_block": g(lo, "level0_ms_per_block_of_1000_snps"), "l0_err": g(lo, "level0_vs_oracle", "W_max_rel_err"),
                  "l1_qt_s": g(lo, "level1_qt_s_per_trait"), "l1_bt_s": g(lo, "level1_bt_s_per_trait")},
        "step2": {"hard_Mvar_s": g(cases, "hard_calls", "variants_per_s"), "hard_masked": g(cases, "hard_calls_masked_phenos", "variants_per_s"),
                  "dos8": g(cases, "dosages_8bit", "variants_per_s"), "dos8_masked": g(cases, "dosages_8bit_masked_phenos", "variants_per_s"),
                  "bgen_var_s": g(run0, "variants_per_s"), "bgen_block_loop_var_s": g(run0, "variants_per_s_block_loop"),
                  "bgen_lines_identical": "%s/%s" % (g(cb0, "byte_identical"), g(cb0, "result_lines"))},
    }
    errs = [k for k in ("config3_single_gpu", "config4_level1_binary_traits", "loocv_500k", "step2") if isinstance(line.get(k), dict) and
            any(str(kk).startswith("error") for kk in line[k])]
    if errs:
        out["errors_in"] = errs
    return out


def bt_oracle_leg(W0, yraw, offset, mask, cv_sizes, tau, q, g_beta, g_cs, prevalence, quasi_newton):
    """`--bt --oracle-check`: oracle/regenie_step1.py's ridge_logistic_level_1 (Step1_Models.cpp:966-1156: Newton steps on the exact fp64 Hessian,
    every ridge value warm-started from the previous one) for the FIRST fold model of trait q at this run's full sample count, against what the
    library's default route returned for the same chain (rg_bt_options.beta_out / fold_cumsum_out).  Checker only: nothing here is timed as product."""
    from oracle import regenie_step1 as orc
    t0 = time.perf_counter()
    opt = orc.Step1Options(bed="", pheno_file="", bt=True)
    cs, betas, ok = orc.ridge_logistic_level_1(W0, yraw, offset, mask, cv_sizes, tau, opt, folds=[0])
    t_or = time.perf_counter() - t0
    n0 = int(cv_sizes[0])
    b_or = betas[0]                                        # L x R1
    b_gp = np.asarray(g_beta).T                            # [R1][L] -> L x R1
    eta_or = W0[:n0] @ b_or                                # held-out linear predictors (without the offset), n0 x R1
    eta_gp = W0[:n0] @ b_gp
    rel = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(b)))       # noqa: E731
    return {"trait": q, "prevalence": prevalence, "fold": 0, "samples": int(W0.shape[0]), "predictors": int(W0.shape[1]), "oracle_converged": bool(ok),
            "route": "default: " + ("quasi-Newton steps on fp16 Hessians, stored and reused (wgram_bf16.hip)" if quasi_newton else "fp64 Hessians (k_wgram128)"),
            "beta_max_rel_err_per_tau": [rel(b_gp[:, j], b_or[:, j]) for j in range(tau.size)],
            "beta_max_rel_err": max(rel(b_gp[:, j], b_or[:, j]) for j in range(tau.size)),
            "held_out_deviance_max_rel_err": float(np.max(np.abs(np.asarray(g_cs)[5] - cs[5]) / np.abs(cs[5]))),
            "held_out_sums_max_rel_err": float(np.max(np.abs(np.asarray(g_cs) - cs) / np.maximum(np.abs(cs), 1e-300))),
            "prediction_max_rel_err": max(rel(eta_gp[:, j], eta_or[:, j]) for j in range(tau.size)),
            "oracle_s": t_or,
            "note": "max |gpu - oracle| / max |oracle| per ridge value; both sides stop at max |score| < 1e-4 (l1_ridge_tol), so they agree to what that "
                    "tolerance leaves open, not to rounding"}


def measured_traffic(dom, nblocks, n_batches, P):
    """HBM bytes per launch (group) of the dominant kernel from the rocprofv3 --pmc passes of THIS build: tools/collect_profiles.sh
    writes profiles/*traffic*.json with the digests of the sources it was measured on (regenie_amd/lib/build.stamp = kernel library + host
    driver, lib/library.stamp = the kernel library alone).  A file measured on other KERNEL sources is refused (traffic = null) rather than
    rescaled; a change to the host driver (regenie_amd/host) does not change what a kernel moves and leaves the files valid, and so does a
    change to a source file that neither defines nor launches the group's kernels (`source_digests`: sha256 per file of the library)."""
    import glob
    try:
        stamp = open(os.path.join(ROOT, "regenie_amd", "lib", "build.stamp")).read().strip()
    except OSError:
        return None, "no build stamp"
    try:
        lib_stamp = open(os.path.join(ROOT, "regenie_amd", "lib", "library.stamp")).read().strip()
    except OSError:
        lib_stamp = None
    group = {"chol_f64": "chol", "l1_gram_f64": "l1_gram", "gram_fp4": "gram_fp4", "pred": "pred", "wgram_f64": "wgram", "irls_stream": "irls_stream"}[dom]
    # the sources that define and launch the group's kernels: a traffic file stays valid for a group while THESE are what it was measured on
    common = ["csrc/rg_api.hip", "csrc/rg_internal.h", "csrc/bed_prep.hip", "flags"]
    group_files = {"chol": ["csrc/chol.hip", "csrc/chol_p128.h", "csrc/chol_common.h", "csrc/assemble.hip"], "l1_gram": ["csrc/l1.hip"], "gram_fp4": ["csrc/gram_fp4.hip"], "pred": ["csrc/pred.hip", "csrc/pred_i8.hip"],
                   "wgram": ["csrc/wgram_bf16.hip", "csrc/l1x.hip"], "irls_stream": ["csrc/l1x.hip"]}[group] + common
    try:
        now = json.load(open(os.path.join(ROOT, "regenie_amd", "lib", "kernel_files.json")))
    except (OSError, ValueError):
        try:      # (a library built before build() wrote the file: the sources beside it are what it was built from, or its stamp would differ)
            from regenie_amd import build as _b
            now = _b.kernel_file_digests() if open(os.path.join(ROOT, "regenie_amd", "lib", "library.stamp")).read().strip() == _b.library_digest() else None
        except Exception:   # noqa: BLE001
            now = None

    def same_group_sources(tj):
        then = tj.get("source_digests")
        return bool(now and then) and all(f in now and now.get(f) == then.get(f) for f in group_files)
    stale, same_build = False, False
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json")), reverse=True):
        try:
            tj = json.load(open(fn))
        except Exception:   # noqa: BLE001
            continue
        if tj.get("build_stamp") != stamp and (lib_stamp is None or tj.get("library_stamp") != lib_stamp) and not same_group_sources(tj):
            stale = True
            continue
        same_build = True
        g = tj.get("groups", {}).get(group)
        if not g or tj.get("blocks") != nblocks or tj.get("phenos") != P:
            continue
        if group == "irls_stream":      # a launch = one streaming pass (k_bt_eval or k_bt_score)
            per = g["hbm_bytes"] / max(1, g.get("group_launches", 0))
        else:
            per = g["hbm_bytes"] / max(1, tj["level0_batches"] if group not in ("l1_gram", "wgram") else g.get("lead_launches", P))
        return per, ("FETCH_SIZE x 2 + WRITE_SIZE of the kernel (group) from separate rocprofv3 --pmc passes of this command on these "
                     "kernel sources (%s), per launch" % os.path.basename(fn))
    if same_build:
        return None, "the PMC traffic file of this build covers another workload (blocks / phenotypes); none was collected for this one"
    return None, ("the committed PMC traffic files were measured on other kernel sources (stale): refused" if stale else
                  "no PMC traffic file for this build / workload (tools/collect_profiles.sh)")


def math_sqrt(x):
    return float(np.sqrt(x))


def from_disk_leg(args, torch, packed, blocks, my_blocks, Yraw, cov, M, N, P, gpu_loco, free_gpu=None, nruns=3):
    """The same workload END TO END through the C++ driver (`regenie-amd --step 1`): the synthetic .bed/.bim/.fam and the
    phenotype / covariate text files are written once to the local disk, then the driver is timed from process start to
    the last .loco byte -- text parsing, context set-up, streamed ingest (reader thread -> page-locked buffers -> PCIe),
    level 0, level 1, LOCO formatting.  BASELINE's metric is a complete --step 1 run; `value` of the main line is the
    resident-data figure, this is the from-files figure next to it."""
    import shutil
    import subprocess
    import tempfile
    drv = os.path.join(ROOT, "regenie_amd", "bin", "regenie-amd")
    if not os.path.exists(drv):
        return None
    need = int(sum(blocks[b][2] for b in my_blocks)) * (N // 4) * 1.15 + 2e9
    free = shutil.disk_usage(tempfile.gettempdir()).free
    if free < need:
        return {"skipped": "the from-files leg needs %.0f GB in %s, %.0f GB are free" % (need / 1e9, tempfile.gettempdir(), free / 1e9)}
    d = tempfile.mkdtemp(prefix="rg_from_disk_")
    try:
        pre = os.path.join(d, "g")
        t0 = time.perf_counter()
        nbytes = 3
        with open(pre + ".bed", "wb") as fh:
            fh.write(b"\x6c\x1b\x01")
            for b in my_blocks:
                buf = packed[b].cpu().numpy()
                nbytes += buf.size
                fh.write(buf.tobytes())
        with open(pre + ".bim", "w") as fh:
            j = 0
            out = []
            for b in my_blocks:
                c = blocks[b][0] + 1
                for _ in range(blocks[b][2]):
                    out.append("%d\ts%d\t0\t%d\tA\tG\n" % (c, j, j + 1))
                    j += 1
            fh.write("".join(out))
        with open(pre + ".fam", "w") as fh:
            fh.write("".join("%d %d 0 0 0 -9\n" % (i + 1, i + 1) for i in range(N)))
        with open(pre + ".pheno", "w") as fh:
            fh.write("FID IID " + " ".join("Y%d" % (p + 1) for p in range(P)) + "\n")
            fh.write("".join("%d %d " % (i + 1, i + 1) + " ".join("%.17g" % v for v in Yraw[i]) + "\n" for i in range(N)))
        with open(pre + ".covar", "w") as fh:
            fh.write("FID IID C1 C2\n")
            fh.write("".join("%d %d %.17g %.17g\n" % (i + 1, i + 1, cov[i, 0], cov[i, 1]) for i in range(N)))
        t_write = time.perf_counter() - t0
        order = sorted(range(N), key=lambda i: "%d_%d" % (i + 1, i + 1))
        got = np.asarray(gpu_loco[0])[order, :].T.copy()
        if free_gpu is not None:                # a workload that fills the device: this process lets go of it before the driver starts
            free_gpu()
        cmd = [drv, "--step", "1", "--bed", pre, "--phenoFile", pre + ".pheno", "--covarFile", pre + ".covar", "--bsize", str(args.bsize),
               "--qt", "--out", os.path.join(d, "o")]
        io = host_io_context(pre + ".bed") if nbytes > (4 << 30) else None      # the large configurations: where the figure is an I/O number
        walls = []
        # A process that releases device memory leaves it to the runtime's scrub (20 - 35 GB/s on this platform, tools/alloc_probe2.cpp:
        # 100 GB allocate in 0.4 ms on a clean device and in 3.3 - 6.9 s right after 100 GB were freed): a driver that starts right behind
        # this process' engine (or behind its own previous run: 170 GB at configs[2]) waits for that, which no run on an idle device does.
        # The large configuration therefore lets the device settle before every run after the first; the first one is reported as it is.
        # (the small configuration too, since round 6: a driver run started right behind the previous one waited 0.3 - 0.4 s for the few GB that one had released)
        settle_s = 15.0 if nbytes > (4 << 30) else 5.0
        for k in range(nruns):                  # first run warms the page cache and the driver's code objects
            if k > 0 and settle_s:
                time.sleep(settle_s)
            t0 = time.perf_counter()
            r = subprocess.run(cmd, capture_output=True, text=True, cwd=d)
            walls.append(time.perf_counter() - t0)
            if r.returncode != 0:
                return {"error": (r.stdout + r.stderr)[-1500:]}
        wall = min(walls[1:]) if len(walls) > 1 else walls[0]
        ids, ref = _parse_loco(os.path.join(d, "o_1.loco"))
        err = float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
        stages = [ln.strip() for ln in r.stdout.splitlines() if "level 0 ridge of blocks" in ln or "-level 1 for" in ln or "Elapsed time" in ln
                  or "since start" in ln]
        rec_io = None
        if io:
            rates = [v for v in list(io["pread_GBps_by_threads"].values()) if v]
            best = max(rates) if rates else None
            rec_io = dict(io, read_ceiling_s=(nbytes / 1e9 / best) if best else None, wall_minus_read_ceiling_s=(wall - nbytes / 1e9 / best) if best else None,
                          page_cache_resident_fraction_after_runs=host_io_context(pre + ".bed", sweep_bytes=0)["page_cache_resident_fraction"])
        return {"wall_s": wall, "value": M * N * P / wall, "unit": "SNP*sample*pheno/s", "walls_s": walls, "bed_bytes": nbytes,
                "bed_GBps": nbytes / wall / 1e9, "host_io": rec_io, "driver_log": stages, "setup_write_s": t_write,
                "loco_text_vs_resident_run_max_rel_err": err, "settle_s_before_runs_after_the_first": settle_s,
                "note": "regenie-amd --step 1 from files on the local disk (page cache warm), process start to exit; best of the runs after the first (a single run: that run)"
                        + ("; walls_s[0] started right behind the release of this process' device memory (the runtime scrubs freed memory at 20 - 35 GB/s and "
                           "allocations wait for it), the later runs on a settled device" if settle_s else "")}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def host_io_context(path, sweep_bytes=8 << 30):
    """What the from-files figure has to be read against: how much of the file the page cache holds right now (mincore over a mapping), what
    this box delivers for plain reads of the same file (pread of 16 MB pieces into per-thread buffers, 4 / 8 / 16 threads; through the page
    cache and with O_DIRECT), and the memory / CPU the container may use.  Bounded: the sweeps stop after `sweep_bytes`."""
    import ctypes
    import mmap
    import threading
    out = {}
    size = os.path.getsize(path)

    def resident():
        fd = os.open(path, os.O_RDONLY)
        try:
            m = mmap.mmap(fd, size, access=mmap.ACCESS_COPY)
            libc = ctypes.CDLL("libc.so.6", use_errno=True)
            npages = (size + 4095) // 4096
            vec = (ctypes.c_ubyte * npages)()
            addr = ctypes.addressof(ctypes.c_char.from_buffer(m))
            rc = libc.mincore(ctypes.c_void_p(addr), ctypes.c_size_t(size), vec)
            frac = float((np.frombuffer(vec, dtype=np.uint8) & 1).mean()) if rc == 0 else None
            del addr
            m.close()
            return frac
        except Exception:   # noqa: BLE001
            return None
        finally:
            os.close(fd)

    def sweep(nthreads, direct):
        flags = os.O_RDONLY | (os.O_DIRECT if direct else 0)
        try:
            fd = os.open(path, flags)
        except OSError:
            return None
        piece = 16 << 20
        total = min(size, sweep_bytes) // piece * piece
        if total == 0:
            os.close(fd)
            return None
        nxt = [0]
        lock = threading.Lock()
        ok = [True]

        def work():
            buf = mmap.mmap(-1, piece)          # page-aligned (O_DIRECT needs it)
            while True:
                with lock:
                    off = nxt[0]
                    nxt[0] += piece
                if off >= total:
                    break
                try:
                    got = os.preadv(fd, [buf], off)
                except OSError:
                    ok[0] = False
                    break
                if got != piece:
                    ok[0] = False
                    break
        t0 = time.perf_counter()
        ths = [threading.Thread(target=work) for _ in range(nthreads)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        os.close(fd)
        return total / dt / 1e9 if ok[0] else None
    out["page_cache_resident_fraction"] = resident()
    out["pread_GBps_by_threads"] = {str(n): sweep(n, False) for n in (4, 8, 16)}
    out["pread_o_direct_GBps_by_threads"] = {str(n): sweep(n, True) for n in (4, 16)}
    out["page_cache_resident_fraction_after_sweeps"] = resident()
    mem = {}
    try:
        for ln in open("/proc/meminfo"):
            k, v = ln.split(":")
            if k in ("MemTotal", "MemAvailable", "Cached"):
                mem[k + "_GB"] = round(int(v.split()[0]) / 1e6, 1)
    except OSError:
        pass
    for k, f in (("cgroup_memory_max", "/sys/fs/cgroup/memory.max"), ("cgroup_cpu_max", "/sys/fs/cgroup/cpu.max")):
        try:
            mem[k] = open(f).read().strip()
        except OSError:
            mem[k] = None
    mem["hardware_threads"] = os.cpu_count()
    out["host"] = mem
    out["sweep_bytes"] = int(min(size, sweep_bytes))
    return out


def _parse_loco(path):
    lines = open(path).read().splitlines()
    ids = lines[0].split()[1:]
    vals = np.array([[float(t) for t in ln.split()[1:]] for ln in lines[1:]])
    return ids, vals


def cpu_baseline(args, eng, torch, dev, packed, blocks, my_blocks, X, Y, Yraw, cov, mask, ain, neff, cv_sizes, tau, M, N, P, B, R0,
                 gpu_full, with_reference=True):
    """CPU leg (rank 0, N=1 only), two parts.

    (1) `kind: "reference"` -- regenie v4.1.2 ITSELF (oracle/_ref/regenie: the reference's sources compiled by
        oracle/Makefile with its own flags) timed on this box's host cores on a BOUNDED sample of the same workload:
        every SNP block of the last two chromosomes (4 blocks, two of them ragged chromosome ends) written to a .bed on
        the local disk with the full sample count and all phenotypes, run as a complete `regenie --step 1` (file parsing,
        level 0, level 1, .loco writing).  The GPU then solves that same sub-problem through the C ABI and the LOCO
        predictors are compared with the reference's files: `loco_max_rel_err` is BASELINE.json's accuracy metric,
        measured against the real reference (its .loco text carries 6 significant digits).
    (2) the numpy oracle (pinned to the reference by tests/test_reference_pin.py) checks the FULL configuration: level-0
        predictors of a sample of blocks, and level 1 of phenotype 0 on the full W -- CV sums, selected ridge value,
        LOCO predictors (`full_config_vs_oracle`)."""
    import shutil
    import subprocess
    import tempfile
    from oracle import regenie_step1 as orc          # timed CPU baseline / checker only
    from regenie_amd import hostprep as hp
    from regenie_amd.engine import Step1Engine
    out = {}
    regenie = os.path.join(ROOT, "oracle", "_ref", "regenie")
    chr_ids = sorted({blocks[b][0] for b in my_blocks})
    sel_chr = chr_ids[-2:]
    sel = [b for b in my_blocks if blocks[b][0] in sel_chr]
    if len(sel) > args.cpu_blocks:
        sel = sel[:max(1, args.cpu_blocks // 2)] + sel[-(args.cpu_blocks - max(1, args.cpu_blocks // 2)):]
    Ms = sum(blocks[b][2] for b in sel)
    if with_reference and os.path.exists(regenie):
        d = tempfile.mkdtemp(prefix="rg_cpu_baseline_")
        try:
            pre = os.path.join(d, "s")
            with open(pre + ".bed", "wb") as fh:
                fh.write(b"\x6c\x1b\x01")
                for b in sel:
                    fh.write(packed[b].cpu().numpy().tobytes())
            with open(pre + ".bim", "w") as fh:
                j = 0
                for b in sel:
                    for _ in range(blocks[b][2]):
                        fh.write("%d\ts%d\t0\t%d\tA\tG\n" % (blocks[b][0] + 1, j, j + 1))
                        j += 1
            with open(pre + ".fam", "w") as fh:
                fh.write("".join("%d %d 0 0 0 -9\n" % (i + 1, i + 1) for i in range(N)))
            with open(pre + ".pheno", "w") as fh:
                fh.write("FID IID " + " ".join("Y%d" % (p + 1) for p in range(P)) + "\n")
                for i in range(N):
                    fh.write("%d %d " % (i + 1, i + 1) + " ".join("%.17g" % v for v in Yraw[i]) + "\n")
            with open(pre + ".covar", "w") as fh:
                fh.write("FID IID C1 C2\n")
                for i in range(N):
                    fh.write("%d %d %.17g %.17g\n" % (i + 1, i + 1, cov[i, 0], cov[i, 1]))
            ncore = os.cpu_count() or 1
            # the reference's own default is all cores - 1 (Regenie.cpp:1104-1106), but its Eigen GEMM does not scale to hundreds
            # of threads: on the 256-thread GPU box two 50,000 x 1,000 blocks take 7.6 s with 16 threads, 9.0 s with 32, 15 s
            # with 64 and > 80 s with 255 (tools/ref_threads_probe.py, profiles/r2_reference_threads.md) -- it gets 16
            thr = max(1, min(ncore - 1, int(os.environ.get("RG_REF_THREADS", "16"))))
            t0 = time.perf_counter()
            r = subprocess.run([regenie, "--step", "1", "--bed", pre, "--phenoFile", pre + ".pheno", "--covarFile", pre + ".covar",
                                "--bsize", str(args.bsize), "--qt", "--threads", str(thr), "--out", os.path.join(d, "ref")],
                               capture_output=True, text=True, cwd=d)
            t_ref = time.perf_counter() - t0
            if r.returncode != 0:
                raise RuntimeError("reference run failed: " + r.stdout[-2000:] + r.stderr[-2000:])
            phases = {"geno_resid_ms": 0.0, "working_matrices_ms": 0.0, "level0_ridge_ms": 0.0}
            for ln in r.stdout.splitlines():
                for key, tag in (("geno_resid_ms", "-residualizing and scaling genotypes...done ("),
                                 ("working_matrices_ms", "-calc working matrices...done ("),
                                 ("level0_ridge_ms", "-calc level 0 ridge...done (")):
                    if tag in ln:
                        phases[key] += float(ln.split(tag)[1].split("ms")[0])
            # the same sub-problem through the C ABI
            sblocks = [(blocks[b][0], None, blocks[b][2]) for b in sel]
            Bs = len(sel)
            h0 = hp.set_ridge_params(R0)
            lam_s = Ms * (1 - h0) / h0
            h1 = hp.set_ridge_params(tau.shape[1])
            Ls = Bs * R0
            tau_s = np.tile(Ls * (1 - h1) / h1, (P, 1))
            e2 = Step1Engine(dev.index or 0, torch.cuda.current_stream().cuda_stream)
            e2.set_problem(X=X, Y=Y, mask=mask, ind_in_analysis=ain, cv_sizes=cv_sizes, lam=lam_s, neff=neff, n_file=N,
                           n_blocks_total=Bs, max_block_size=args.bsize)
            e2.l0_blocks_device(list(range(Bs)), [blocks[b][2] for b in sel], [packed[b].data_ptr() for b in sel], N // 4)
            e2.sync()
            cols = [sum(R0 for b in sel if blocks[b][0] == c) for c in sel_chr]
            e2.set_loco_output([c + 1 for c in sel_chr])
            cs2, best2, pred2 = e2.l1_qt(tau_s, cols)
            err, tab_ok = 0.0, True
            order = sorted(range(N), key=lambda i: "%d_%d" % (i + 1, i + 1))      # the writer's std::map order (Data.cpp:1934)
            for p in range(P):
                ids, ref = _parse_loco(os.path.join(d, "ref_%d.loco" % (p + 1)))
                got = np.asarray(pred2[p])[order, :].T
                err = max(err, float(np.max(np.abs(got - ref)) / np.max(np.abs(ref))))
            mins = [ln for ln in r.stdout.splitlines() if "<- min value" in ln]
            h1s = ["%g" % v for v in h1]
            ref_best = [h1s.index(ln.split(":")[0].strip()) if ln.split(":")[0].strip() in h1s else -1 for ln in mins]
            e2.close()
            out.update({"value": Ms * N * P / t_ref, "unit": "SNP*sample*pheno/s", "cores": thr, "kind": "reference",
                        "sample": "regenie v4.1.2 (oracle/_ref/regenie, -O3 -ffast-math -fopenmp, Eigen 3.4.0, --threads %d of %d host "
                                  "cores): complete --step 1 from files on %d of the run's %d SNP blocks (all blocks of chromosomes %s: "
                                  "%d SNPs x %d samples x %d phenotypes), %.1f s wall" % (thr, ncore, Bs, B, [c + 1 for c in sel_chr], Ms, N, P, t_ref),
                        "sample_wall_s": t_ref, "reference_phase_ms": phases,
                        "extrapolated_total_s": (phases["geno_resid_ms"] + phases["working_matrices_ms"] + phases["level0_ridge_ms"]) * 1e-3 / Ms * M,
                        "extrapolation_note": "level-0 phases of the reference's own log scaled linearly in SNPs to the full run (level 1 and I/O not included)",
                        "loco_max_rel_err": err, "selected_tau_index_reference": ref_best,
                        "selected_tau_index_gpu": [int(b) for b in best2]})
        finally:
            shutil.rmtree(d, ignore_errors=True)
    # ---- the numpy oracle on the full configuration ----
    prep = orc.Prepared(ids=[], n_file=N, ind_ignore=np.zeros(N, bool), ind_in_analysis=ain, pheno_names=[],
                        Y=Y, Y_raw=None, mask=mask, X=X, Neff=neff, scale_Y=np.ones(P), ncov=X.shape[1],
                        n_analyzed=N)
    lam = M * (1 - hp.set_ridge_params(R0)) / hp.set_ridge_params(R0)
    nsel = min(2, len(my_blocks))
    osel = [my_blocks[int(i)] for i in np.linspace(0, len(my_blocks) - 1, nsel)]
    t_l0, werr = 0.0, 0.0
    for b in osel:
        rows = packed[b].cpu().numpy()
        t0 = time.perf_counter()
        G = orc.read_chunk_from_bed(rows, N, None, ain)
        G, _ = orc.residualize_genotypes(G, prep)
        Wb = orc.ridge_level_0(G, prep, cv_sizes, lam)
        t_l0 += time.perf_counter() - t0
        for p in range(P):
            werr = max(werr, float(np.max(np.abs(eng.get_w(b, p) - Wb[p])) / np.max(np.abs(Wb[p]))))
    W0 = np.concatenate([eng.get_w(b, 0) for b in range(B)], axis=1)
    t0 = time.perf_counter()
    cs, betas = orc.ridge_level_1(W0, Y[:, 0], cv_sizes, tau[0])
    best = orc.select_tau(cs, neff[0], False)
    chrcols = orc.chr_columns([(c + 1, s0, n) for (c, s0, n) in blocks], sorted({bl[0] + 1 for bl in blocks}), R0)
    pred = orc.make_predictions(W0, betas, best, cv_sizes, chrcols)
    loco = orc.loco_from_predictions(pred, chrcols, 23)
    t_l1 = time.perf_counter() - t0
    g_loco, g_cs, g_best = gpu_full
    full = {"W_max_rel_err_on_%d_blocks" % nsel: werr,
            # Sx / Sy are sums of centred values (~1e-11): errors are taken relative to the largest entry of the table
            "l1_cumsum_max_err_rel_to_max": float(np.max(np.abs(np.asarray(g_cs)[0][:5] - cs[:5])) / np.max(np.abs(cs[:5]))),
            "selected_tau_index_oracle": int(best), "selected_tau_index_gpu": int(g_best[0]),
            "loco_max_rel_err_pheno0": float(np.max(np.abs(np.asarray(g_loco[0]) - loco)) / np.max(np.abs(loco))),
            "oracle_s": {"level0_%d_blocks" % nsel: t_l0, "level1_pheno0": t_l1}}
    out["full_config_vs_oracle"] = full
    if "value" not in out:       # no reference binary on this box: the numpy restatement is the baseline
        total = t_l0 / nsel * B + t_l1 * P
        out.update({"value": M * N * P / total, "unit": "SNP*sample*pheno/s", "cores": os.cpu_count(), "kind": "port",
                    "sample": "oracle (numpy+OpenBLAS fp64): level 0 on %d of %d blocks (%.1f s) extrapolated linearly + level 1 of one "
                              "phenotype on the full W (%.1f s) x P" % (nsel, B, t_l0, t_l1), "extrapolated_total_s": total,
                    "loco_max_rel_err": full["loco_max_rel_err_pheno0"]})
    return out


if __name__ == "__main__":
    main()
