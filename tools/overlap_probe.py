"""Probe: do two independent level-0 pipelines (two contexts, two streams, disjoint block ranges, one shared W)
overlap on the GPU?  Compares one context over all blocks with two contexts over the two halves."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from regenie_amd import hostprep as hp
from regenie_amd.engine import Step1Engine
dev = torch.device("cuda", 0)
N, P, bsize, M = 50000, 1, 1000, 100000
spc = bench.snps_per_chrom(M); blocks = hp.chrom_blocks(spc, bsize); B = len(blocks); R0 = 5
packed = {b: bench.gen_block(torch, dev, b, blocks[b][2], N, 1234, 10)[0] for b in range(B)}
rng = np.random.default_rng(99); cov = rng.standard_normal((N, 2)); Yraw = rng.standard_normal((N, P))
X = hp.get_basis(np.concatenate([np.ones((N, 1)), cov], axis=1)); mask = np.ones((N, P), bool); neff = np.full(P, float(N))
Y, _ = hp.residualize_pheno(Yraw - Yraw.mean(axis=0), X, mask, neff); ain = np.ones(N, bool); cv = hp.set_folds(ain, 5)
lam = M * (1 - hp.set_ridge_params(R0)) / hp.set_ridge_params(R0)
def mk(stream):
    e = Step1Engine(0, stream.cuda_stream)
    e.set_problem(X=X, Y=Y, mask=mask, ind_in_analysis=ain, cv_sizes=cv, lam=lam, neff=neff, n_file=N, n_blocks_total=B, max_block_size=bsize)
    return e
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
e0, e1 = mk(s0), mk(s1)
Wt = torch.zeros(e0.w_bytes // 8, dtype=torch.float64, device=dev)
e0.set_w_buffer(Wt.data_ptr(), e0.w_bytes); e1.set_w_buffer(Wt.data_ptr(), e1.w_bytes)
ptrs = [packed[b].data_ptr() for b in range(B)]; bss = [blocks[b][2] for b in range(B)]
h = (B + 1) // 2
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e0.l0_blocks_device(list(range(B)), bss, ptrs, N // 4); e0.sync()
    t1 = time.perf_counter()
    e0.l0_blocks_device(list(range(h)), bss[:h], ptrs[:h], N // 4)
    e1.l0_blocks_device(list(range(h, B)), bss[h:], ptrs[h:], N // 4)
    e0.sync(); e1.sync(); t2 = time.perf_counter()
    print("one context %.1f ms   two contexts/streams %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
