"""The multi-GPU hand-off of the level-0 predictors (regenie_amd/csrc/rg_group.hip: rg_group_create / rg_group_prepare / rg_l0_finish and the
all-reduce it installs for the shared level 1) EXECUTED WITH 2 - 4 RANKS in this container: the source is compiled for the host against
tests/hipcpu (device memory = host memory, streams synchronous), the ranks are host threads as in the product, and RCCL is a stand-in
(tests/hipcpu/fake_rccl.cpp, named to the library through RG_RCCL_LIB) that matches sends and receives by (source, destination) in issue order and aborts
on a count that differs between the two ends.  What this holds: who sends which rows of W to whom, packed how, received where -- the block
ranges and phenotype ranges uneven, a rank without blocks, W held for the rank's own range only -- for BOTH transports (RCCL: never run on more
than one device so far; peer copies: run with two contexts on one device in tests/test_distributed_gpu.py), the all-gather form, and the
all-reduce callback.  What it does not: RCCL itself, xGMI, the overlap with level 0."""
import ctypes as C
import os
import subprocess
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RCCL, PEER = 0, 1


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("grpemu")
    hc = os.path.join(ROOT, "tests", "hipcpu")
    fake, so = str(d / "librccl_fake.so"), str(d / "libgrpemu.so")
    for cmd in (["g++", "-O1", "-std=c++17", "-fPIC", "-w", "-shared", "-Wl,-soname,librccl.so.1", os.path.join(hc, "fake_rccl.cpp"), "-o", fake, "-lpthread"],
                ["g++", "-O1", "-std=c++17", "-fPIC", "-w", "-shared", "-x", "c++", "-I" + hc, os.path.join(ROOT, "regenie_amd", "csrc", "rg_group.hip"),
                 os.path.join(hc, "group_host.cpp"), "-o", so, "-ldl", "-lpthread"]):
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
    os.environ["RG_RCCL_LIB"] = fake                 # (rg_group.hip resolves RCCL at run time; this names the build it loads)
    rccl = C.CDLL(fake, mode=C.RTLD_GLOBAL)
    L = C.CDLL(so)
    L.emu_ctx_create.restype = C.c_void_p
    L.emu_ctx_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int]
    L.emu_ctx_destroy.argtypes = [C.c_void_p]
    L.emu_ctx_w.restype = C.POINTER(C.c_double)
    L.emu_ctx_w.argtypes = [C.c_void_p]
    L.emu_ctx_error.restype = C.c_char_p
    L.emu_ctx_error.argtypes = [C.c_void_p]
    L.emu_ctx_view.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.emu_ctx_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.emu_blocks_done.argtypes = [C.c_void_p]
    L.rg_group_create.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p), C.c_int]
    L.rg_group_destroy.argtypes = [C.c_void_p]
    L.rg_group_destroy.restype = None
    L.rg_l0_finish.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.rg_group_prepare.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.rg_group_abort.argtypes = [C.c_void_p, C.c_int32]
    L.rg_group_abort.restype = None
    L._rccl = rccl
    return L


def _global_w(B, R0, P, Np, seed):
    return np.random.default_rng(seed).standard_normal((B * R0, P, Np))


def _run_ranks(n, fn):
    out, th = [None] * n, []
    for r in range(n):
        def work(r=r):
            out[r] = fn(r)
        th.append(threading.Thread(target=work))
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    assert not any(t.is_alive() for t in th), "a rank is still waiting in the exchange"
    return out


CASES = [  # ranks, block ranges, phenotype ranges, ranged W
    (2, [0, 3, 5], [0, 2, 3], False),
    (2, [0, 3, 5], [0, 2, 3], True),
    (3, [0, 2, 2, 7], [0, 1, 4, 5], True),            # a rank without blocks
    (4, [0, 1, 4, 6, 9], [0, 3, 4, 6, 7], True),
    (4, [0, 3, 5, 6, 8], [0, 1, 2, 3, 4], False),
]


@pytest.mark.parametrize("transport", [RCCL, PEER])
@pytest.mark.parametrize("n,bb,pb,ranged", CASES)
def test_exchange_by_phenotype(lib, transport, n, bb, pb, ranged):
    R0, Np = 3, 256
    B, P = bb[-1], pb[-1]
    Wg = _global_w(B, R0, P, Np, 7)
    ctxs = []
    for r in range(n):
        b0, nb = (bb[r], bb[r + 1] - bb[r]) if ranged else (0, B)
        c = lib.emu_ctx_create(r, R0, P, Np, B, b0, nb)
        w = np.ctypeslib.as_array(lib.emu_ctx_w(c), shape=(max(nb, 0) * R0, P, Np)) if nb > 0 else None
        if w is not None:
            w[...] = np.nan                                   # rows of the others' blocks are never read
            lo = (bb[r] - b0) * R0
            w[lo:lo + (bb[r + 1] - bb[r]) * R0] = Wg[bb[r] * R0:bb[r + 1] * R0]
        ctxs.append(c)
    g = C.c_void_p()
    arr = (C.c_void_p * n)(*ctxs)
    assert lib.rg_group_create(C.byref(g), n, arr, transport) == 0, lib.emu_ctx_error(ctxs[0])
    cbb, cpb = (C.c_int32 * (n + 1))(*bb), (C.c_int32 * (n + 1))(*pb)
    assert lib.rg_group_prepare(g, 0, cbb, cpb) == 0          # (one rank allocates ahead, the others inside rg_l0_finish)
    rcs = _run_ranks(n, lambda r: lib.rg_l0_finish(g, r, cbb, cpb))
    assert rcs == [0] * n, [lib.emu_ctx_error(c) for c in ctxs]
    for r in range(n):
        wv, p0, npn, world, rank = C.POINTER(C.c_double)(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        lib.emu_ctx_view(ctxs[r], C.byref(wv), C.byref(p0), C.byref(npn), C.byref(world), C.byref(rank))
        assert (p0.value, npn.value, world.value) == (pb[r], pb[r + 1] - pb[r], 1)          # level 1 of its phenotypes, alone
        view = np.ctypeslib.as_array(wv, shape=(B * R0, npn.value, Np))
        assert np.array_equal(view, Wg[:, pb[r]:pb[r + 1], :]), "rank %d" % r
    if transport == RCCL:
        st = (C.c_int64 * 4)()
        lib._rccl.fake_rccl_stats(st)
        want = sum((bb[r + 1] - bb[r]) * R0 * (pb[k + 1] - pb[k]) * Np * 8 for r in range(n) for k in range(n) if k != r)
        assert st[1] == want                                   # every row crosses once, to the one rank that needs it (1 / n of the all-gather volume)
    lib.rg_group_destroy(g)
    for c in ctxs:
        lib.emu_ctx_destroy(c)


@pytest.mark.parametrize("transport", [RCCL, PEER])
@pytest.mark.parametrize("n,bb", [(2, [0, 3, 5]), (3, [0, 2, 2, 7]), (4, [0, 1, 4, 6, 9])])
def test_all_gather_and_shared_all_reduce(lib, transport, n, bb):
    R0, Np, P = 2, 128, 2
    B = bb[-1]
    Wg = _global_w(B, R0, P, Np, 11)
    ctxs = []
    for r in range(n):
        c = lib.emu_ctx_create(r, R0, P, Np, B, 0, B)
        w = np.ctypeslib.as_array(lib.emu_ctx_w(c), shape=(B * R0, P, Np))
        w[...] = -777.0
        w[bb[r] * R0:bb[r + 1] * R0] = Wg[bb[r] * R0:bb[r + 1] * R0]
        ctxs.append(c)
    g = C.c_void_p()
    assert lib.rg_group_create(C.byref(g), n, (C.c_void_p * n)(*ctxs), transport) == 0
    cbb = (C.c_int32 * (n + 1))(*bb)
    rcs = _run_ranks(n, lambda r: lib.rg_l0_finish(g, r, cbb, None))
    assert rcs == [0] * n, [lib.emu_ctx_error(c) for c in ctxs]
    for r in range(n):
        assert np.array_equal(np.ctypeslib.as_array(lib.emu_ctx_w(ctxs[r]), shape=(B * R0, P, Np)), Wg)
        assert lib.emu_blocks_done(ctxs[r]) == B
        wv, p0, npn, world, rank = C.POINTER(C.c_double)(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        lib.emu_ctx_view(ctxs[r], C.byref(wv), C.byref(p0), C.byref(npn), C.byref(world), C.byref(rank))
        assert not wv and (p0.value, npn.value, world.value, rank.value) == (0, P, n, r)      # every phenotype, level 1 shared among the ranks
    # the all-reduce the shared level 1 calls between its kernels: twice, lengths that differ
    for length in (1000, 17):
        bufs = [np.random.default_rng(100 + r).standard_normal(length) for r in range(n)]
        want = np.sum(bufs, axis=0)
        rcs = _run_ranks(n, lambda r: lib.emu_ctx_allreduce(ctxs[r], bufs[r].ctypes.data, length))
        assert rcs == [0] * n
        for r in range(n):
            assert np.allclose(bufs[r], want, rtol=1e-15, atol=1e-15) and np.array_equal(bufs[r], bufs[0])     # the same bits on every rank
    lib.rg_group_destroy(g)
    for c in ctxs:
        lib.emu_ctx_destroy(c)


@pytest.mark.parametrize("transport", [RCCL, PEER])
def test_a_failed_rank_releases_the_others(lib, transport):
    """rg_group_abort instead of rg_l0_finish on one rank: the others come back with an error instead of waiting for it."""
    n, bb, pb, R0, Np = 3, [0, 2, 4, 6], [0, 1, 2, 3], 2, 128
    ctxs = [lib.emu_ctx_create(r, R0, 3, Np, 6, 0, 6) for r in range(n)]
    g = C.c_void_p()
    assert lib.rg_group_create(C.byref(g), n, (C.c_void_p * n)(*ctxs), transport) == 0
    cbb, cpb = (C.c_int32 * (n + 1))(*bb), (C.c_int32 * (n + 1))(*pb)

    def rank(r):
        if r == 1:
            lib.rg_group_abort(g, 1)
            return -1
        return lib.rg_l0_finish(g, r, cbb, cpb)
    rcs = _run_ranks(n, rank)
    assert rcs[0] != 0 and rcs[2] != 0
    assert b"another GPU" in lib.emu_ctx_error(ctxs[0])
    lib.rg_group_destroy(g)
    for c in ctxs:
        lib.emu_ctx_destroy(c)
