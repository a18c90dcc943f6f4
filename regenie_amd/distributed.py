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

    W: torch tensor [B*R0 (+ slack rows), P, Np] (float64); on every rank the columns of its own blocks are filled.
    Equal slabs: one all_gather_into_tensor straight into W.  Uneven slabs (B not divisible by the world size:
    the block counts differ by one): ONE all-gather of equal-size, zero-padded slabs into a staging tensor followed by
    device-side copies of the valid rows -- RCCL drives all xGMI links at once, where a sequence of per-rank
    broadcasts would serialise `world` transfers.  force_broadcast keeps the broadcast sequence (debug / comparison).
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1:
        return W
    rank = dist.get_rank(group)
    sizes = {nb for (_, nb) in shards}
    nmax = max(nb for (_, nb) in shards)
    row_bytes = W[0].numel() * W.element_size()
    stage_too_big = world * nmax * r0 * row_bytes > (16 << 30)   # a second copy of W must fit beside W (config 3: it does not)
    if force_broadcast or (len(sizes) > 1 and stage_too_big):
        for src, (b0, nb) in enumerate(shards):
            if nb == 0:
                continue
            dist.broadcast(W[b0 * r0:(b0 + nb) * r0], src=src, group=group)
        return W
    if len(sizes) == 1 and shards[0][1] > 0:
        nb = shards[0][1]
        mine = W[shards[rank][0] * r0:(shards[rank][0] + nb) * r0]
        dist.all_gather_into_tensor(W[: world * nb * r0], mine.clone(), group=group)
        return W
    b0, nb = shards[rank]
    mine = torch.zeros((nmax * r0,) + tuple(W.shape[1:]), dtype=W.dtype, device=W.device)
    mine[: nb * r0] = W[b0 * r0:(b0 + nb) * r0]
    stage = torch.empty((world * nmax * r0,) + tuple(W.shape[1:]), dtype=W.dtype, device=W.device)
    dist.all_gather_into_tensor(stage, mine, group=group)
    for src, (s0, sn) in enumerate(shards):
        if sn > 0 and src != rank:
            W[s0 * r0:(s0 + sn) * r0] = stage[src * nmax * r0: src * nmax * r0 + sn * r0]
    return W


def shard_phenotypes(n_pheno: int, world: int) -> List[Tuple[int, int]]:
    """[(first_phenotype, count)] per rank, contiguous and balanced (same rule as shard_blocks)."""
    return shard_blocks(n_pheno, world)


def exchange_w_by_phenotype(W, shards: List[Tuple[int, int]], pshards: List[Tuple[int, int]], r0: int, group=None,
                            via_host: bool = False, buffers: dict = None, own_rows_only: bool = False):
    """Phenotype-sharded hand-off of the level-0 predictors: ONE all-to-all in which rank r sends to rank g
    the predictor rows of r's blocks for g's phenotypes only -- 1/world of the all-gather volume -- and every rank ends
    up with W_g [L, count_g, Np], the layout the library's level-1 view expects (rg_set_l1_view).

    W: [B*R0, P, Np] with this rank's block columns filled, or (own_rows_only, rg_set_block_range) just the rows of this
    rank's blocks [nb*R0, P, Np].  Returns the received tensor (None if this rank owns no phenotype).  `buffers`: a dict
    the send / receive tensors are kept in across calls (one strided copy per destination into a preallocated buffer
    instead of fresh temporaries: at BASELINE configs[2] the send side alone is tens of GB).  via_host stages the exchange
    through host memory (gloo test mode)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    b0, nb = shards[rank]
    P, Np = W.shape[1], W.shape[2]
    mine = W[: nb * r0] if own_rows_only else W[b0 * r0:(b0 + nb) * r0]   # [nb*R0, P, Np]
    in_split = [nb * r0 * qn * Np for (_, qn) in pshards]
    q0, qn = pshards[rank]
    out_split = [sn * r0 * qn * Np for (_, sn) in shards]
    if buffers is None:
        buffers = {}
    key = (sum(in_split), sum(out_split), W.dtype, W.device)     # the cached pair fits this exchange exactly, or is replaced
    if buffers.get("key") != key:
        buffers["send"] = torch.empty(sum(in_split), dtype=W.dtype, device=W.device)
        buffers["recv"] = torch.empty(sum(out_split), dtype=W.dtype, device=W.device)
        buffers["key"] = key
    send, recv = buffers["send"], buffers["recv"]
    off = 0
    for (p0, pn), cnt in zip(pshards, in_split):
        if cnt:
            send[off:off + cnt].view(nb * r0, pn, Np).copy_(mine[:, p0:p0 + pn, :])
        off += cnt
    if via_host:
        hs, hr = send.cpu(), torch.empty(sum(out_split), dtype=W.dtype)
        dist.all_to_all_single(hr, hs, output_split_sizes=out_split, input_split_sizes=in_split, group=group)
        recv.copy_(hr)
    else:
        dist.all_to_all_single(recv, send, output_split_sizes=out_split, input_split_sizes=in_split, group=group)
    if qn == 0:
        return None
    L = sum(sn for (_, sn) in shards) * r0
    return recv.view(L, qn, Np)
