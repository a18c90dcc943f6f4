#!/usr/bin/env python
"""Step-1 benchmark (BASELINE.json metric: Step-1 SNPs x samples x phenos / sec).

A "step" is one complete pass of the Step-1 hot path over the synthetic workload with the packed
genotypes already resident in HBM: level 0 over every SNP block (decode/impute, FP4 matrix-core fold
Gram, fp64 multi-lambda ridge solves, out-of-fold predictions, standardisation), [N>1: all-gather of
the level-0 predictors], level 1 (fold Grams, K*R1 ridge solves, CV statistics, tau selection,
per-chromosome predictions; N>1: Gram tiles and ridge systems shared among the ranks, two all-reduces)
and the LOCO assembly on the host.

N=1 workload = BASELINE.json configs[1]: synthetic PLINK bed, 50K samples x 100K SNPs, 1 QT phenotype,
bsize 1000, 22 chromosomes.  N>1: weak scaling, every rank processes its own 100K SNPs (M = 100K * N).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--samples N] [--snps M] [--phenos P] [--no-cpu]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# hg38 autosome lengths (Mb), used only to spread SNPs over 22 chromosomes
CHR_MB = [248, 242, 198, 190, 181, 171, 159, 145, 138, 133, 135, 133, 114, 107, 102, 90, 83, 80, 59, 64, 47, 51]

PEAK = {"fp4_mfma_TOPS": 10000.0,    # MX FP4 dense (MI355X_MICROARCH.md: ~10 PF dense, ubench 9099 TF at 32x32x64)
        "i8_mfma_TOPS": 5000.0,      # 2x the bf16 dense peak (MI355X_MICROARCH.md: I8 ~2x bf16 rate; ubench 4404)
        "f64_mfma_TFLOPS": 78.6,     # AMD datasheet FP64 matrix (not listed in the guide)
        "hbm_GBs": 8000.0}


def snps_per_chrom(M):
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
    ap.add_argument("--samples", type=int, default=50000)
    ap.add_argument("--snps", type=int, default=100000, help="SNPs per GPU")
    ap.add_argument("--phenos", type=int, default=1)
    ap.add_argument("--bsize", type=int, default=1000)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-blocks", type=int, default=4)
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

    N, P, bsize = args.samples, args.phenos, args.bsize
    assert N % 4 == 0
    M = args.snps * world                                   # weak scaling: 100K SNPs per GPU
    spc = snps_per_chrom(M)
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
    Y, _ = hp.residualize_pheno(Yraw - Yraw.mean(axis=0), X, mask, neff)
    ain = np.ones(N, bool)
    cv_sizes = hp.set_folds(ain, 5)
    lam = M * (1 - hp.set_ridge_params(R0)) / hp.set_ridge_params(R0)
    L = B * R0
    h1 = hp.set_ridge_params(R1)
    tau = np.tile(L * (1 - h1) / h1, (P, 1))
    cols_per_chr = [sum(1 for bl in blocks if bl[0] == c) * R0 for c in range(len(spc))]
    chroms = [c + 1 for c, n in enumerate(cols_per_chr) if n > 0]
    cols_per_chr = [n for n in cols_per_chr if n > 0]
    t_gen = time.time() - t_gen

    eng = Step1Engine(local, torch.cuda.current_stream().cuda_stream)
    eng.set_problem(X=X, Y=Y, mask=mask, ind_in_analysis=ain, cv_sizes=cv_sizes, lam=lam, neff=neff,
                    n_file=N, n_blocks_total=B, max_block_size=bsize)
    Wt = torch.zeros(eng.w_bytes // 8, dtype=torch.float64, device=dev)
    eng.set_w_buffer(Wt.data_ptr(), eng.w_bytes)
    Wv = Wt.view(L, P, eng.w_rows)
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
    pheno_sharded = world > 1 and P >= world
    pshards = shard_phenotypes(P, world) if pheno_sharded else None
    if world > 1 and not pheno_sharded:
        eng.set_collective(world, rank, _allreduce)

    def step(exchange=True, solo=False):
        eng.l0_blocks_device(my_blocks, bss, ptrs, N // 4)
        eng.sync()
        if pheno_sharded and not solo:
            Wg = exchange_w_by_phenotype(Wv, shards, pshards, R0, via_host=(args.backend != "nccl"))
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
        if rank == 0 or (world > 1 and not solo):           # N>1: level 1 is shared among the ranks
            cs, best, pred = eng.l1_qt(tau, cols_per_chr)
            out = [pred[p] for p in range(P)], cs, best                                    # LOCO rows, assembled on the device
        return out

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

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
            "chol_f64": sum((x ** 3 / 3.0 + 2.0 * x * x * P) * 5 * R0 for x in bss),  # K*R0 systems per block
            "l1_gram_f64": 2.0 * N * L * L * P,
            "l1_chol_f64": P * 5 * R1 * (L ** 3 / 3.0 + 2.0 * L * L),
        }
        kernels = {
            "gram_fp4": {"ms": tm["ms_gram"], "achieved_TOPS": flops["gram_fp4"] / (tm["ms_gram"] * 1e-3) / 1e12 if tm["ms_gram"] else None,
                        "launches": tm["n_gram_launches"]},
            "chol_f64": {"ms": tm["ms_chol"], "achieved_TFLOPS": flops["chol_f64"] / (tm["ms_chol"] * 1e-3) / 1e12 if tm["ms_chol"] else None},
            "prep": {"ms": tm["ms_prep"]}, "geno_xy": {"ms": tm["ms_xy"]}, "assemble_form": {"ms": tm["ms_assemble"]},
            "pred": {"ms": tm["ms_pred"]},
            "l1_gram_f64": {"ms": tm["ms_l1_gram"], "achieved_TFLOPS": flops["l1_gram_f64"] / (tm["ms_l1_gram"] * 1e-3) / 1e12 if tm["ms_l1_gram"] else None},
            "l1_chol_f64": {"ms": tm["ms_l1_chol"]}, "l1_cv_pred": {"ms": tm["ms_l1_pred"]},
        }
        dom = max(("gram_fp4", "chol_f64", "l1_gram_f64"), key=lambda k: kernels[k]["ms"])
        kernels["gram_fp4"]["frac_of_fp4_peak"] = kernels["gram_fp4"]["achieved_TOPS"] / PEAK["fp4_mfma_TOPS"]
        if dom == "gram_fp4":
            a = kernels[dom]["achieved_TOPS"]
            roof = {"kernel": "k_gram_fp4_blocks (FP4 matrix-core fold Gram)", "bound": "mfma", "achieved": a, "peak": PEAK["fp4_mfma_TOPS"],
                    "unit": "TOP/s", "frac": a / PEAK["fp4_mfma_TOPS"], "traffic": None,
                    "algorithmic_ops_per_launch": flops["gram_fp4"] / max(1, n_batches), "avg_launch_ms": tm["ms_gram"] / max(1, n_batches)}
        else:
            a = kernels[dom]["achieved_TFLOPS"]
            traffic = None
            if dom == "chol_f64":   # HBM bytes per batch from the separate rocprofv3 PMC passes of this same command
                try:                # (profiles/r1_traffic.json, tools/pmc_traffic.py: FETCH_SIZE x 2 + WRITE_SIZE)
                    tj = json.load(open(os.path.join(ROOT, "profiles", "r1_traffic.json")))
                    traffic = tj["hbm_bytes_per_batch"] * (len(my_blocks) / n_batches) / (109 / 2.0)
                except Exception:   # noqa: BLE001 - the field is optional
                    traffic = None
            roof = {"kernel": {"chol_f64": "k_chol_update/gfact/gstrip/backsolve (fp64 MFMA batched Cholesky, per level-0 batch of systems)",
                               "l1_gram_f64": "k_l1_gram (fp64 MFMA fold Gram)"}[dom], "bound": "mfma", "achieved": a,
                    "peak": PEAK["f64_mfma_TFLOPS"], "unit": "TFLOP/s", "frac": a / PEAK["f64_mfma_TFLOPS"], "traffic": traffic,
                    "traffic_note": "HBM bytes per launch group from separate rocprofv3 --pmc passes (profiles/r1_traffic.json), scaled by blocks per batch",
                    "algorithmic_flops_per_launch": flops[dom] / max(1, n_batches if dom == "chol_f64" else P),
                    "avg_launch_ms": kernels[dom]["ms"] / max(1, n_batches if dom == "chol_f64" else P)}

    # ---- CPU baseline: the oracle (numpy/OpenBLAS restatement of the reference) on a bounded sample ----
    cpu = None
    if rank == 0 and not args.no_cpu and world == 1:
        cpu = cpu_baseline(args, eng, packed, blocks, my_blocks, X, Y, mask, ain, neff, cv_sizes, lam, tau, M, N, P, B, R0)

    if rank == 0:
        line = {
            "metric": "Step-1 SNPs x samples x phenos / sec", "value": value, "unit": "SNP*sample*pheno/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": sec_per_step * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp4 (exact integer Gram) + f64 (solves, level 1)",
            "data": "synthetic",
            "config": {"workload": "synthetic PLINK bed %d samples x %d SNPs (%d per GPU), %d QT pheno, bsize %d, 22 chromosomes, 5-fold CV, 5x5 ridge grid"
                       % (N, M, args.snps, P, bsize), "samples": N, "snps": M, "phenos": P, "bsize": bsize, "blocks": B,
                       "parallelism": ("blocks sharded x%d, all-to-all of W by phenotype, level 1 phenotype-sharded x%d" if pheno_sharded else
                                       "blocks sharded x%d, all-gather of W, level-1 Gram tiles / ridge systems shared x%d") % (world, world)},
            "roofline": roof, "kernels": kernels, "cpu_baseline": cpu,
            "loco_checksum": float(res[3]) if len(res) > 3 else float(sum(np.abs(l).sum() for l in res[0])),
            "selected_tau_index": [int(b) for b in res[2]],
            "setup_s": {"generate": t_gen},
        }
        print(json.dumps(line))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


def math_sqrt(x):
    return float(np.sqrt(x))


def cpu_baseline(args, eng, packed, blocks, my_blocks, X, Y, mask, ain, neff, cv_sizes, lam, tau, M, N, P, B, R0):
    """Times the oracle on the GPU box's host cores: level 0 on `cpu_blocks` full blocks (level 0 is exactly
    linear in the number of blocks) + level 1 for one phenotype on the full W, extrapolated to the run."""
    from oracle import regenie_step1 as orc          # timed CPU baseline / checker only
    prep = orc.Prepared(ids=[], n_file=N, ind_ignore=np.zeros(N, bool), ind_in_analysis=ain, pheno_names=[],
                        Y=Y, Y_raw=None, mask=mask, X=X, Neff=neff, scale_Y=np.ones(P), ncov=X.shape[1],
                        n_analyzed=N)
    nsel = min(args.cpu_blocks, len(my_blocks))
    sel = [my_blocks[int(i)] for i in np.linspace(0, len(my_blocks) - 1, nsel)]
    t_l0, err = 0.0, 0.0
    for b in sel:
        rows = packed[b].cpu().numpy()
        t0 = time.perf_counter()
        G = orc.read_chunk_from_bed(rows, N, None, ain)
        G, _ = orc.residualize_genotypes(G, prep)
        Wb = orc.ridge_level_0(G, prep, cv_sizes, lam)
        t_l0 += time.perf_counter() - t0
        for p in range(P):
            err = max(err, float(np.max(np.abs(eng.get_w(b, p) - Wb[p])) / np.max(np.abs(Wb[p]))))
    W0 = np.concatenate([eng.get_w(b, 0) for b in range(B)], axis=1)
    t0 = time.perf_counter()
    cs, betas = orc.ridge_level_1(W0, Y[:, 0], cv_sizes, tau[0])
    t_l1 = time.perf_counter() - t0
    total = t_l0 / nsel * B + t_l1 * P
    return {"value": M * N * P / total, "unit": "SNP*sample*pheno/s", "cores": os.cpu_count(), "kind": "port",
            "sample": "oracle (numpy+OpenBLAS fp64, all host threads): level 0 on %d of %d blocks (%.1f s) extrapolated linearly in "
                      "blocks + level 1 of one phenotype on the full W (%.1f s) x P" % (nsel, B, t_l0, t_l1),
            "extrapolated_total_s": total, "gpu_vs_oracle_W_max_rel_err_on_sample": err}


if __name__ == "__main__":
    main()
