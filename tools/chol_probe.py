#!/usr/bin/env python
"""Times the batched Cholesky (rg_k_chol_solve) on a level-0-sized batch: `batch` systems of order n with one RHS row
tile.  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split; RG_CHOL_DBG / RG_CHOL_V select variants."""
import sys
import time

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from regenie_amd.engine import load_library


def main(n=1024, batch=1375, reps=3):
    lib = load_library()
    dev = "cuda"
    T = n // 64
    rtot = n + 64
    g = torch.Generator(device=dev).manual_seed(1)
    base = torch.randn(n, 2 * n, dtype=torch.float64, device=dev, generator=g)
    A = torch.tril(base @ base.T + 50.0 * torch.eye(n, dtype=torch.float64, device=dev))
    src = torch.zeros(rtot, n, dtype=torch.float64, device=dev)
    src[:n] = A
    src[n:n + 1] = torch.randn(1, n, dtype=torch.float64, device=dev, generator=g)
    mats = torch.empty(batch, rtot, n, dtype=torch.float64, device=dev)
    dinv = torch.zeros(batch * (T + 10 * ((T + 3) // 4)) * 4096, dtype=torch.float64, device=dev)
    info = torch.zeros(4, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for rep in range(reps):
        mats.copy_(src.unsqueeze(0).expand(batch, rtot, n))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = lib.rg_k_chol_solve(st, mats.data_ptr(), rtot * n, batch, n, 64, 1, dinv.data_ptr(), info.data_ptr())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert rc == 0
        print("rep %d: %.3f ms" % (rep, dt * 1e3))
    x = mats[0, n, :].cpu()
    L = torch.linalg.cholesky((A + torch.tril(A, -1).T).cpu())
    ref = torch.cholesky_solve(src[n:n + 1].cpu().T, L).T[0]
    print("max rel err", float((x - ref).abs().max() / ref.abs().max()))


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:]])
