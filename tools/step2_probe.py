"""Device time of the Step-2 QT score-test kernels (include/rg_step2.h) on synthetic dosages resident in HBM.
Prints one JSON line per configuration: ms per block, variants/s, and the achieved rate against the algorithmic bytes
(one 8-byte read of every genotype; the current two-pass kernels read the block twice)."""
import json
import os
import sys

import numpy as np
import torch

from regenie_amd.step2 import Step2QT


def run(n, C, P, bs, reps=5):
    rng = np.random.default_rng(0)
    X = np.linalg.qr(np.column_stack([np.ones(n), rng.normal(size=(n, C - 1))]))[0]
    mask = (rng.random((P, n)) > 0.05).astype(np.uint8)
    yres = rng.normal(size=(P, n)) * mask
    G = torch.bernoulli(torch.full((bs, n), 0.3, dtype=torch.float64, device="cuda")) * 2.0
    G += torch.rand_like(G) * 0.01
    torch.cuda.synchronize()
    with Step2QT(n, C, P) as s2:
        s2.set_null(X.T, yres, mask, np.ones(P))
        ms = []
        for _ in range(reps):
            ms.append(s2.score_block(G)["kernel_ms"])
    best, med = min(ms), sorted(ms)[len(ms) // 2]
    alg = bs * n * 8.0
    print(json.dumps({"tile": os.environ.get("RG_S2_TILE", "default"), "n": n, "C": C, "P": P, "bs": bs, "ms_median": round(med, 4), "ms_best": round(best, 4),
                      "variants_per_s": round(bs / med * 1e3), "algorithmic_GBps": round(alg / med / 1e6, 1),
                      "frac_of_8TBps": round(alg / med / 1e6 / 8000, 4)}), flush=True)


if __name__ == "__main__":
    cfgs = [(100_000, 10, 10, 512), (200_000, 10, 10, 512), (500_000, 10, 10, 256), (500_000, 20, 50, 256)]
    if len(sys.argv) > 1:
        cfgs = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
    tiles = os.environ.get("RG_S2_TILES", "").split(",") if os.environ.get("RG_S2_TILES") else [None]
    for tile in tiles:
        if tile:
            os.environ["RG_S2_TILE"] = tile
        for c in cfgs:
            run(*c)
