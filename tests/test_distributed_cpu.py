"""N>1 path on CPU: block sharding (the reference's --split-l0 job split, src/Data.cpp:270-302) and the
all-gather of level-0 predictor slabs, world size 2 over gloo."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

from regenie_amd.distributed import allgather_w, shard_blocks  # noqa: E402


def test_shard_blocks_matches_split_l0_rule():
    assert shard_blocks(10, 4) == [(0, 3), (3, 3), (6, 2), (8, 2)]        # floor(B/n), first B mod n get one more
    assert shard_blocks(8, 8) == [(i, 1) for i in range(8)]
    assert shard_blocks(3, 4) == [(0, 1), (1, 1), (2, 1), (3, 0)]
    for B in (1, 7, 109, 522):
        for n in (1, 2, 4, 8):
            s = shard_blocks(B, n)
            assert sum(nb for _, nb in s) == B and all(s[i][0] + s[i][1] == s[i + 1][0] for i in range(n - 1))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, R0, P, Np, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shards = shard_blocks(B, world)
    W = torch.zeros(B * R0, P, Np, dtype=torch.float64)
    b0, nb = shards[rank]
    full = torch.arange(B * R0 * P * Np, dtype=torch.float64).view(B * R0, P, Np)
    W[b0 * R0:(b0 + nb) * R0] = full[b0 * R0:(b0 + nb) * R0]             # each rank fills only its own columns
    allgather_w(W, shards, R0)
    q.put((rank, bool(torch.equal(W, full))))
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [6, 7])                                    # even and uneven slabs
def test_allgather_w_gloo_world2(B):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, 5, 2, 64, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def _worker_a2a(rank, world, port, B, R0, P, Np, q):
    from regenie_amd.distributed import exchange_w_by_phenotype, shard_phenotypes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shards = shard_blocks(B, world)
    pshards = shard_phenotypes(P, world)
    full = torch.arange(B * R0 * P * Np, dtype=torch.float64).view(B * R0, P, Np)
    W = torch.zeros_like(full)
    b0, nb = shards[rank]
    W[b0 * R0:(b0 + nb) * R0] = full[b0 * R0:(b0 + nb) * R0]             # each rank fills only its own block columns
    Wg = exchange_w_by_phenotype(W, shards, pshards, R0)
    q0, qn = pshards[rank]
    ok = bool(torch.equal(Wg, full[:, q0:q0 + qn, :])) and tuple(Wg.shape) == (B * R0, qn, Np)
    # a rank that keeps the rows of its own blocks only (rg_set_block_range), buffers reused across calls
    bufs = {}
    own = full[b0 * R0:(b0 + nb) * R0].clone()
    for _ in range(2):
        Wg2 = exchange_w_by_phenotype(own, shards, pshards, R0, buffers=bufs, own_rows_only=True)
        ok = ok and bool(torch.equal(Wg2, full[:, q0:q0 + qn, :]))
    q.put((rank, ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("B,P", [(6, 4), (7, 3)])                       # even / uneven block and phenotype shards
def test_exchange_w_by_phenotype_gloo_world2(B, P):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_a2a, args=(r, 2, port, B, 5, P, 64, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]
