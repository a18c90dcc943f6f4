#!/usr/bin/env python
"""Times the fp64 level-0 path (rg_l0_blocks_f64) on config-2-sized blocks: N samples, blocks of 1000 dosage variants held in
device memory.  Usage: python tools/f64_probe.py [N=50000] [nblocks=8]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctypes as C

import numpy as np
import torch

from regenie_amd import hostprep as hp
from regenie_amd.engine import RG_MEM_DEVICE, Step1Engine


def main(N=50000, nb=8, bs=1000):
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(1)
    P = 1
    X = hp.get_basis(np.concatenate([np.ones((N, 1)), rng.standard_normal((N, 2))], axis=1))
    mask = np.ones((N, P), bool)
    neff = np.full(P, float(N))
    Y, _ = hp.residualize_pheno(rng.standard_normal((N, P)), X, mask, neff)
    ain = np.ones(N, bool)
    cv = hp.set_folds(ain, 5)
    lam = nb * bs * (1 - hp.set_ridge_params(5)) / hp.set_ridge_params(5)
    eng = Step1Engine(0)
    eng.set_problem(X=X, Y=Y, mask=mask, ind_in_analysis=ain, cv_sizes=cv, lam=lam, neff=neff, n_file=N, n_blocks_total=nb, max_block_size=bs)
    g = torch.Generator(device=dev).manual_seed(3)
    maf = torch.rand(bs, 1, device=dev, generator=g) * 0.4 + 0.05
    blocks = []
    for b in range(nb):
        d = (torch.rand(bs, N, device=dev, generator=g) < maf).double() + (torch.rand(bs, N, device=dev, generator=g) < maf).double()
        d = torch.clamp(d + 0.1 * torch.randn(bs, N, device=dev, generator=g, dtype=torch.float64), 0, 2)
        blocks.append(torch.round(d * 16384) / 16384)
    ids = np.arange(nb, dtype=np.int32)
    bss = np.full(nb, bs, dtype=np.int32)
    ptrs = (C.c_void_p * nb)(*[int(t.data_ptr()) for t in blocks])
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = eng.lib.rg_l0_blocks_f64(eng.h, nb, ids.ctypes.data, bss.ctypes.data, ptrs, N, RG_MEM_DEVICE)
        assert rc == 0
        eng.sync()
        dt = time.perf_counter() - t0
        flops = nb * (2.0 * N * 1024 * 1024 + 25 * bs ** 3 / 3.0)
        print("rep %d: %.1f ms for %d blocks = %.2f ms/block, %.1f TFLOP/s (Gram + solves), %.2e SNP*sample/s" % (
            rep, dt * 1e3, nb, dt * 1e3 / nb, flops / dt / 1e12, nb * bs * N / dt))
    eng.close()


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    main(*a)
