"""Per-kernel parity (GPU): each hand-written HIP kernel is called through its C-ABI test entry and
compared with numpy on the same seeded inputs.  Integer kernels: bit-exact.  fp64 kernels: 1e-11."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from regenie_amd.engine import load_library  # noqa: E402
from tests.util import pack_bed, synth_dosages  # noqa: E402


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("m,n,nsamp,miss", [(128, 128, 256, 0.0), (200, 130, 1024, 0.02), (1000, 1000, 4096, 0.01),
                                            (37, 300, 64, 0.3)])
def test_gram_i8_exact(m, n, nsamp, miss):
    lib = load_library()
    ga = synth_dosages(m, nsamp, miss, seed=1)
    gb = synth_dosages(n, nsamp, miss, seed=2)
    pa, pb = pack_bed(ga), pack_bed(gb)                      # nsamp multiple of 64 -> rows of nsamp/4 bytes
    A, B = _dev(pa), _dev(pb)
    for a_miss in (0, 1):
        for b_miss in (0, 1):
            Cd = torch.full((m, n), -7, dtype=torch.int32, device="cuda")
            rc = lib.rg_k_gram_i8(_stream(), A.data_ptr(), pa.shape[1], a_miss, B.data_ptr(), pb.shape[1], b_miss,
                                  m, n, pa.shape[1], Cd.data_ptr(), n)
            assert rc == 0
            torch.cuda.synchronize()
            fa = (ga == -3).astype(np.int64) if a_miss else np.where(ga < 0, 0, ga).astype(np.int64)
            fb = (gb == -3).astype(np.int64) if b_miss else np.where(gb < 0, 0, gb).astype(np.int64)
            ref = fa @ fb.T
            assert np.array_equal(Cd.cpu().numpy().astype(np.int64), ref), (a_miss, b_miss)


def test_gram_i8_asymmetric_layout():
    """A=I-style check with an asymmetric B so a row/col swap of the MFMA C/D map cannot pass."""
    lib = load_library()
    m = n = 128
    nsamp = 128
    ga = np.zeros((m, nsamp), np.int8)
    ga[np.arange(m), np.arange(m)] = 1                        # A = identity (dosage 1 on the diagonal)
    gb = ((np.arange(n)[:, None] * 3 + np.arange(nsamp)[None, :]) % 3).astype(np.int8)
    pa, pb = pack_bed(ga), pack_bed(gb)
    A, B = _dev(pa), _dev(pb)
    Cd = torch.zeros((m, n), dtype=torch.int32, device="cuda")
    assert lib.rg_k_gram_i8(_stream(), A.data_ptr(), 32, 0, B.data_ptr(), 32, 0, m, n, 32, Cd.data_ptr(), n) == 0
    torch.cuda.synchronize()
    assert np.array_equal(Cd.cpu().numpy(), ga.astype(np.int32) @ gb.astype(np.int32).T)


@pytest.mark.parametrize("m,n,k", [(64, 64, 64), (128, 192, 640), (256, 64, 4096),
                                   # the LDS-staged kernel (k_dgemm_nt128): whole macro tiles, 64-row remainders in both dimensions, many super tiles,
                                   # more than eight macro-tile rows (two super-tile rows per XCD pass); the entry point takes K in multiples of 64
                                   (128, 128, 64), (192, 320, 1088), (1024, 8192, 1024), (2688, 1280, 2688), (1152, 4160, 192)])
def test_dgemm_nt(m, n, k):
    lib = load_library()
    rng = np.random.default_rng(5)
    A = rng.standard_normal((m, k))
    B = rng.standard_normal((n, k))
    Ad, Bd = _dev(A), _dev(B)
    Cd = torch.zeros((m, n), dtype=torch.float64, device="cuda")
    assert lib.rg_k_dgemm_nt(_stream(), Ad.data_ptr(), k, Bd.data_ptr(), k, m, n, k, Cd.data_ptr(), n) == 0
    torch.cuda.synchronize()
    ref = A @ B.T
    assert np.max(np.abs(Cd.cpu().numpy() - ref)) <= 1e-11 * np.max(np.abs(ref))


@pytest.mark.parametrize("n,nrhs,batch,rhs_pad", [(64, 1, 3, 64), (100, 2, 2, 64), (200, 5, 4, 64), (1000, 3, 2, 64),
                                                  (300, 2, 3, 64), (545, 70, 2, 128), (256, 130, 9, 192),
                                                  (130, 300, 2, 320)])
@pytest.mark.parametrize("path", ["group", "column"])
def test_chol_solve(n, nrhs, batch, rhs_pad, path, monkeypatch):
    """tile counts 1..16 (partial last column group: 5, 9 tiles), several right-hand-side row tiles (odd totals); both
    factorization paths: group-wise (large batches) and per-column (small batches)"""
    monkeypatch.setenv("RG_CHOL_SMALL", "0" if path == "group" else "1000000")
    lib = load_library()
    rng = np.random.default_rng(11)
    n64 = (n + 63) // 64 * 64
    rtot = n64 + rhs_pad
    mats = np.zeros((batch, rtot, n64))
    sols = []
    for b in range(batch):
        G = rng.standard_normal((n, 3 * n))
        A = G @ G.T + (10.0 + b) * np.eye(n)
        rhs = rng.standard_normal((nrhs, n))
        mats[b, :n, :n] = np.tril(A)
        mats[b, np.arange(n, n64), np.arange(n, n64)] = 1.0
        mats[b, n64:n64 + nrhs, :n] = rhs
        sols.append(np.linalg.solve(A, rhs.T).T)
    Md = _dev(mats)
    T = n64 // 64
    dinv = torch.zeros(batch * (T + 10 * ((T + 3) // 4)) * 4096, dtype=torch.float64, device="cuda")
    info = torch.zeros(4, dtype=torch.int32, device="cuda")
    rc = lib.rg_k_chol_solve(_stream(), Md.data_ptr(), rtot * n64, batch, n64, rhs_pad, nrhs, dinv.data_ptr(), info.data_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    out = Md.cpu().numpy()
    assert int(info[0].item()) == 0
    for b in range(batch):
        x = out[b, n64:n64 + nrhs, :n]
        assert np.max(np.abs(x - sols[b])) <= 1e-10 * np.max(np.abs(sols[b])), b
        Lg = np.tril(out[b, :n, :n])
        Lr = np.linalg.cholesky(np.tril(mats[b, :n, :n]) + np.tril(mats[b, :n, :n], -1).T)
        assert np.max(np.abs(Lg - Lr)) <= 1e-10 * np.max(np.abs(Lr))


def test_chol_flags_non_spd():
    lib = load_library()
    n64, rtot = 64, 128
    mats = np.zeros((1, rtot, n64))
    mats[0, :64, :64] = -np.eye(64)
    Md = _dev(mats)
    dinv = torch.zeros(11 * 4096, dtype=torch.float64, device="cuda")
    info = torch.zeros(4, dtype=torch.int32, device="cuda")
    assert lib.rg_k_chol_solve(_stream(), Md.data_ptr(), rtot * n64, 1, n64, 64, 1, dinv.data_ptr(), info.data_ptr()) == 0
    torch.cuda.synchronize()
    assert int(info[0].item()) != 0


def _pack_fp4(g):
    """int dosage (0/1/2, negative = missing -> 0) -> FP4 E2M1 nibbles (1 -> 0b0010, 2 -> 0b0100), two per byte."""
    nib = np.zeros(g.shape, np.uint8)
    nib[g == 1] = 0x2
    nib[g == 2] = 0x4
    return (nib[:, 0::2] | (nib[:, 1::2] << 4)).astype(np.uint8)


@pytest.mark.parametrize("m,n,nsamp", [(256, 256, 256), (300, 260, 1024), (1000, 1000, 4096), (37, 513, 512)])
def test_gram_fp4_exact(m, n, nsamp):
    """FP4 matrix-core Gram: bit-exact integers (random dosages, ragged tiles, asymmetric operands)."""
    lib = load_library()
    ga = np.where(synth_dosages(m, nsamp, 0.02, seed=11) < 0, 0, synth_dosages(m, nsamp, 0.02, seed=11))
    gb = np.where(synth_dosages(n, nsamp, 0.0, seed=12) < 0, 0, synth_dosages(n, nsamp, 0.0, seed=12))
    pa, pb = _pack_fp4(ga), _pack_fp4(gb)
    A, B = _dev(pa), _dev(pb)
    Cd = torch.full((m, n), -7, dtype=torch.int32, device="cuda")
    rc = lib.rg_k_gram_fp4(_stream(), A.data_ptr(), pa.shape[1], B.data_ptr(), pb.shape[1], m, n, pa.shape[1],
                           Cd.data_ptr(), n)
    assert rc == 0
    torch.cuda.synchronize()
    assert np.array_equal(Cd.cpu().numpy().astype(np.int64), ga.astype(np.int64) @ gb.astype(np.int64).T)


def test_gram_fp4_identity_asymmetric():
    """A = identity against an asymmetric B: catches a row/col swap of the C/D map and any k mis-pairing."""
    lib = load_library()
    m = n = 256
    nsamp = 256
    ga = np.zeros((m, nsamp), np.int8)
    ga[np.arange(m), np.arange(m)] = 1
    gb = ((np.arange(n)[:, None] * 5 + np.arange(nsamp)[None, :] * 7) % 3).astype(np.int8)
    A, B = _dev(_pack_fp4(ga)), _dev(_pack_fp4(gb))
    Cd = torch.zeros((m, n), dtype=torch.int32, device="cuda")
    assert lib.rg_k_gram_fp4(_stream(), A.data_ptr(), 128, B.data_ptr(), 128, m, n, 128, Cd.data_ptr(), n) == 0
    torch.cuda.synchronize()
    assert np.array_equal(Cd.cpu().numpy(), ga.astype(np.int32) @ gb.astype(np.int32).T)


def test_gram_fp4_large_k_exact():
    """All-2 dosages over 2.5M samples: every entry is 4*K = 10,485,760 -- beyond one fp32-exact super-chunk,
    so the K split with integer atomics is exercised; the sum must still be exact."""
    lib = load_library()
    m = n = 64
    nsamp = 5 * (1 << 19)
    pa = np.full((m, nsamp // 2), 0x44, np.uint8)
    A = _dev(pa)
    Cd = torch.zeros((m, n), dtype=torch.int32, device="cuda")
    assert lib.rg_k_gram_fp4(_stream(), A.data_ptr(), pa.shape[1], A.data_ptr(), pa.shape[1], m, n, pa.shape[1],
                             Cd.data_ptr(), n) == 0
    torch.cuda.synchronize()
    assert np.array_equal(Cd.cpu().numpy(), np.full((m, n), 4 * nsamp, np.int32))
