"""Multi-GPU plumbing for Step 1: one process per GPU, blocks sharded like regenie's own
`--split-l0` job split (reference src/Data.cpp:270-302 write_l0_master: floor(B/n) blocks per job, the
first B mod n jobs get one more, contiguous ranges), and ONE exchange step: an all-gather of the
level-0 predictor column slabs (the `_l0_Y*` file hand-off of Step1_Models.cpp:1941-1987 becomes an RCCL
all-gather over xGMI).  torch.distributed is plumbing only; backend "nccl" is RCCL on ROCm, "gloo" is
used by the CPU tests.
"""
from __future__ import annotations

from typing import List, Tuple


def shard_blocks(n_blocks: int, world: int) -> List[Tuple[int, int]]:
    """[(first_block, n_blocks)] per rank -- contiguous, balanced as write_l0_master does."""
    base, extra = divmod(n_blocks, world)
    out, b0 = [], 0
    for r in range(world):
        nb = base + (1 if r < extra else 0)
        out.append((b0, nb))
        b0 += nb
    return out


def allgather_w(W, shards: List[Tuple[int, int]], r0: int, group=None, force_broadcast: bool = False):
    """In-place all-gather of the level-0 predictors.

    W: torch tensor [B*R0, P, Np] (float64), on every rank the columns of its own blocks are filled.
    Each rank broadcasts its contiguous column slab; with uneven slabs this is a sequence of
    broadcasts (RCCL schedules each as a direct xGMI transfer to the 7 peers), equal slabs use one
    all_gather_into_tensor.
    """
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1:
        return W
    rank = dist.get_rank(group)
    sizes = {nb for (_, nb) in shards}
    if len(sizes) == 1 and shards[0][1] > 0 and not force_broadcast:
        nb = shards[0][1]
        mine = W[shards[rank][0] * r0:(shards[rank][0] + nb) * r0]
        dist.all_gather_into_tensor(W[: world * nb * r0], mine.clone(), group=group)
        return W
    for src, (b0, nb) in enumerate(shards):
        if nb == 0:
            continue
        dist.broadcast(W[b0 * r0:(b0 + nb) * r0], src=src, group=group)
    return W
