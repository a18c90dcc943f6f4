"""Probe: logistic-ridge level 1 (BT, K-fold) at BASELINE config-2 size (50,000 samples, 109 blocks -> L = 545)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from regenie_amd import hostprep as hp
from regenie_amd.engine import Step1Engine
dev = torch.device("cuda", 0)
N, P, bsize, M = 50000, 1, 1000, 100000
spc = bench.snps_per_chrom(M); blocks = hp.chrom_blocks(spc, bsize); B = len(blocks); R0 = R1 = 5
packed, yc = {}, torch.zeros(N, dtype=torch.float64, device=dev)
for b in range(B):
    pk, y = bench.gen_block(torch, dev, b, blocks[b][2], N, 1234, 10); packed[b] = pk; yc += y
rng = np.random.default_rng(99); cov = rng.standard_normal((N, 2))
g = yc.cpu().numpy(); g = g / g.std()
liab = np.sqrt(0.3) * g + np.sqrt(0.7) * rng.standard_normal(N) + 0.2 * cov[:, 0]
yraw = (liab > np.quantile(liab, 0.8)).astype(np.float64)[:, None]
X = hp.get_basis(np.concatenate([np.ones((N, 1)), cov], axis=1)); mask = np.ones((N, P), bool); neff = np.full(P, float(N))
# null logistic offsets (covariates only): a few IRLS steps on the host
beta = np.zeros(X.shape[1])
for _ in range(25):
    eta = X @ beta; pr = 1 / (1 + np.exp(-eta)); w = pr * (1 - pr)
    beta = beta + np.linalg.solve((X.T * w) @ X, X.T @ (yraw[:, 0] - pr))
offset = (X @ beta)[:, None]
Y, _ = hp.residualize_pheno(yraw - yraw.mean(axis=0), X, mask, neff); ain = np.ones(N, bool); cv = hp.set_folds(ain, 5)
lam = M * (1 - hp.set_ridge_params(R0)) / hp.set_ridge_params(R0)
eng = Step1Engine(0, torch.cuda.current_stream().cuda_stream)
eng.set_problem(X=X, Y=Y, mask=mask, ind_in_analysis=ain, cv_sizes=cv, lam=lam, neff=neff, n_file=N, n_blocks_total=B, max_block_size=bsize)
ptrs = [packed[b].data_ptr() for b in range(B)]; bss = [blocks[b][2] for b in range(B)]
L = B * R0; h1 = hp.set_ridge_params(R1); tau = np.tile(L * (1 - h1) / h1 * 3 / np.pi ** 2, (P, 1))
cols = [sum(1 for bl in blocks if bl[0] == c) * R0 for c in range(len(spc))]; cols = [n for n in cols if n > 0]
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.l0_blocks_device(list(range(B)), bss, ptrs, N // 4); eng.sync(); t1 = time.perf_counter()
    cs, conv, best, pred = eng.l1_bt(tau, yraw, offset, cols); t2 = time.perf_counter()
    print("level 0 %.1f ms   BT level 1 (K-fold, 5 tau) %.1f ms   converged %s best %s  -logLik/N %s" %
          ((t1 - t0) * 1e3, (t2 - t1) * 1e3, conv, best, np.round(cs[0][5] / N, 5)))
