"""Host-side prerequisites of the Step-1 hot path (tiny N x C / N x P fp64 work that stays on the CPU,
SURVEY.md 8a row a23), for Python hosts of the C ABI (bench.py, examples).  Product code: never touches the oracle package.  Mirrors, with file:line citations into the reference:
  set_ridge_params   src/Regenie.cpp:1497-1508
  get_basis          src/Pheno.cpp:1660-1681 (getBasis)
  residualize_pheno  src/Pheno.cpp:1799-1834 (residualize_phenotypes)
  set_folds          src/Data.cpp:401-426
  chrom_blocks       src/Data.cpp:319-333, :579-586
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import numpy as np


def set_ridge_params(n: int) -> np.ndarray:
    v = np.arange(n, dtype=np.float64) / (n - 1)
    v[0], v[-1] = 0.01, 0.99
    return v


def get_basis(X: np.ndarray, rel_tol: float = 1e-15) -> np.ndarray:
    d, v = np.linalg.eigh(X.T @ X)
    nz = int((d > d[-1] * rel_tol).sum())
    return (X @ v[:, -nz:]) / np.sqrt(d[-nz:])[None, :]


def residualize_pheno(Y: np.ndarray, X: np.ndarray, mask: np.ndarray, neff: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    beta = Y.T @ X
    Y = Y - (X @ beta.T) * mask
    scale = np.linalg.norm(Y, axis=0) / np.sqrt(neff - X.shape[1])
    if scale.min() < 1e-6:
        raise ValueError("phenotype has sd=0.")
    return Y / scale[None, :], scale


def set_folds(ind_in_analysis: np.ndarray, cv_folds: int) -> np.ndarray:
    a = np.asarray(ind_in_analysis, bool)
    N = a.size
    target = int(a.sum()) // cv_folds
    if target < 1:
        raise ValueError("not enough samples are present for %d-fold CV." % cv_folds)
    if a.all():                      # common case: closed form of the loop below
        sizes = np.full(cv_folds, target, np.int64)
        sizes[-1] = N - target * (cv_folds - 1)
        return sizes
    sizes = np.ones(cv_folds, np.int64)
    cnt = cum = cur = 0
    for i in range(N):
        cnt += int(a[i])
        if cnt == target:
            sizes[cur] = i - cum + 1
            cum += sizes[cur]
            cnt = 0
            cur += 1
        elif cur == cv_folds - 1:
            sizes[cur] = N - i
            break
    return sizes


def chrom_blocks(snps_per_chrom: Sequence[int], bsize: int) -> List[Tuple[int, int, int]]:
    """[(chrom_index, first_snp, bs)] in genome order; blocks never span chromosomes."""
    out, pos = [], 0
    for ci, n in enumerate(snps_per_chrom):
        nb = int(math.ceil(n / bsize))
        for bb in range(nb):
            bs = bsize if (bb + 1) * bsize <= n else n - bb * bsize
            out.append((ci, pos + bb * bsize, bs))
        pos += n
    return out
