import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from regenie_amd import hostprep as hp
from regenie_amd.engine import Step1Engine, loco_from_predictions
dev = torch.device("cuda", 0)
N, P, bsize = 50000, 1, 1000
M = 100000
spc = bench.snps_per_chrom(M)
blocks = hp.chrom_blocks(spc, bsize)
B = len(blocks); R0 = R1 = 5
packed = {}
for b in range(B):
    pk, yc = bench.gen_block(torch, dev, b, blocks[b][2], N, 1234, 10)
    packed[b] = pk
rng = np.random.default_rng(99)
cov = rng.standard_normal((N, 2))
Yraw = rng.standard_normal((N, P))
X = hp.get_basis(np.concatenate([np.ones((N, 1)), cov], axis=1))
mask = np.ones((N, P), bool); neff = np.full(P, float(N))
Y, _ = hp.residualize_pheno(Yraw - Yraw.mean(axis=0), X, mask, neff)
ain = np.ones(N, bool); cv_sizes = hp.set_folds(ain, 5)
lam = M * (1 - hp.set_ridge_params(R0)) / hp.set_ridge_params(R0)
L = B * R0; h1 = hp.set_ridge_params(R1); tau = np.tile(L * (1 - h1) / h1, (P, 1))
cols = [sum(1 for bl in blocks if bl[0] == c) * R0 for c in range(len(spc))]
chroms = [c + 1 for c, n in enumerate(cols) if n > 0]; cols = [n for n in cols if n > 0]
eng = Step1Engine(0, torch.cuda.current_stream().cuda_stream)
eng.set_problem(X=X, Y=Y, mask=mask, ind_in_analysis=ain, cv_sizes=cv_sizes, lam=lam, neff=neff, n_file=N, n_blocks_total=B, max_block_size=bsize)
Wt = torch.zeros(eng.w_bytes // 8, dtype=torch.float64, device=dev); eng.set_w_buffer(Wt.data_ptr(), eng.w_bytes)
ptrs = [packed[b].data_ptr() for b in range(B)]; bss = [blocks[b][2] for b in range(B)]
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.l0_blocks_device(list(range(B)), bss, ptrs, N // 4)
    t1 = time.perf_counter()
    eng.sync(); t2 = time.perf_counter()
    cs, best, pred = eng.l1_qt(tau, cols); t3 = time.perf_counter()
    lo = [loco_from_predictions(pred[p], chroms) for p in range(P)]; t4 = time.perf_counter()
    print("l0 launch %.1f  l0 wait %.1f  l1 %.1f  loco %.1f  total %.1f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, (t4-t0)*1e3))
