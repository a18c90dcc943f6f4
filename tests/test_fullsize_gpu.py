"""Size-independent properties at BASELINE.json's config-2 and config-3 sample counts (50,000 and 500,000 samples,
bsize 1000), where the oracle would take minutes to hours: exact-Gram checksum of checksums, symmetry and diagonal; level-0 predictors invariant
under a rescaling of the phenotype (the column standardisation makes W scale-free) and under the batch composition;
level 1 reproduces the phenotype sign flip (LOCO(-y) = -LOCO(y))."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from regenie_amd import hostprep as hp  # noqa: E402
from regenie_amd.engine import Step1Engine, load_library, loco_from_predictions  # noqa: E402

BS = 1000
N = 50000          # module default; the level-0 / level-1 property tests also run at 500,000


def _gen(seed, bs=BS, n=50000):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    maf = 0.05 + 0.45 * torch.rand(bs, 1, generator=g, device="cuda")
    d = (torch.rand(bs, n, generator=g, device="cuda") < maf).to(torch.uint8) + \
        (torch.rand(bs, n, generator=g, device="cuda") < maf).to(torch.uint8)
    return d


def _pack_bed(d):
    code = torch.where(d == 2, torch.zeros_like(d), torch.where(d == 1, torch.full_like(d, 2), torch.full_like(d, 3)))
    c = code.view(d.shape[0], d.shape[1] // 4, 4)
    return (c[:, :, 0] | (c[:, :, 1] << 2) | (c[:, :, 2] << 4) | (c[:, :, 3] << 6)).contiguous()


def test_fp4_gram_checksums_full_size():
    lib = load_library()
    n = 50176                                       # 50,000 rounded up to a multiple of 256 (one LDS stage)
    d = torch.zeros(BS, n, dtype=torch.uint8, device="cuda")
    d[:, :N] = _gen(1, n=N)
    nib = torch.where(d == 1, torch.full_like(d, 2), torch.where(d == 2, torch.full_like(d, 4), torch.zeros_like(d)))
    p4 = (nib[:, 0::2] | (nib[:, 1::2] << 4)).contiguous()
    S = torch.full((BS, BS), -1, dtype=torch.int32, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.rg_k_gram_fp4(st, p4.data_ptr(), p4.shape[1], p4.data_ptr(), p4.shape[1], BS, BS, p4.shape[1],
                             S.data_ptr(), BS) == 0
    torch.cuda.synchronize()
    S64 = S.to(torch.int64)
    d64 = d.to(torch.int64)
    assert torch.equal(S64, S64.T)                                              # symmetry
    assert torch.equal(torch.diagonal(S64), (d64 * d64).sum(dim=1))             # diagonal = sum of squares
    assert int(S64.sum()) == int((d64.sum(dim=0) ** 2).sum())                   # 1' S 1 = sum_s (column sum)^2
    assert torch.equal(S64.sum(dim=1), (d64 * d64.sum(dim=0)[None, :]).sum(dim=1))   # row checksums


def _problem(Yraw, nblk):
    N = Yraw.shape[0]
    rng = np.random.default_rng(5)
    cov = rng.standard_normal((N, 2))
    X = hp.get_basis(np.concatenate([np.ones((N, 1)), cov], axis=1))
    P = Yraw.shape[1]
    mask = np.ones((N, P), bool)
    neff = np.full(P, float(N))
    Y, _ = hp.residualize_pheno(Yraw - Yraw.mean(axis=0), X, mask, neff)
    ain = np.ones(N, bool)
    cv = hp.set_folds(ain, 5)
    h = hp.set_ridge_params(5)
    lam = nblk * BS * (1 - h) / h
    eng = Step1Engine(0)
    eng.set_problem(X=X, Y=Y, mask=mask, ind_in_analysis=ain, cv_sizes=cv, lam=lam, neff=neff, n_file=N,
                    n_blocks_total=nblk, max_block_size=BS)
    return eng


@pytest.mark.parametrize("N", [50000, 500000])
def test_level0_scale_and_batch_invariance_full_size(N):
    nblk = 3
    packed = [_pack_bed(_gen(10 + b, n=N)) for b in range(nblk)]
    rng = np.random.default_rng(1)
    y = rng.standard_normal((N, 1))
    outs = []
    for scale, order in ((1.0, [0, 1, 2]), (-3.5, [2, 0, 1])):
        eng = _problem(scale * y, nblk)
        if scale > 0:
            eng.l0_blocks_device(order, [BS] * nblk, [packed[b].data_ptr() for b in order], N // 4)
        else:                                                    # one block per call, permuted order
            for b in order:
                eng.l0_blocks_device([b], [BS], [packed[b].data_ptr()], N // 4)
        eng.sync()
        outs.append(np.concatenate([eng.get_w(b, 0) for b in range(nblk)], axis=1))
        eng.close()
    W1, W2 = outs
    assert np.all(np.isfinite(W1)) and abs(W1.std() - 1.0) < 1e-3               # standardised columns
    # y -> -3.5 y flips the sign of every prediction column and nothing else; batches of 1 vs 3 blocks are identical
    assert np.max(np.abs(W1 + W2)) < 1e-9 * np.max(np.abs(W1))


@pytest.mark.parametrize("N", [50000, 500000])
def test_level1_sign_flip_full_size(N):
    nblk = 2
    packed = [_pack_bed(_gen(20 + b, n=N)) for b in range(nblk)]
    rng = np.random.default_rng(2)
    y = rng.standard_normal((N, 1))
    locos = []
    for sgn in (1.0, -1.0):
        eng = _problem(sgn * y, nblk)
        eng.l0_blocks_device(list(range(nblk)), [BS] * nblk, [p.data_ptr() for p in packed], N // 4)
        eng.sync()
        L = nblk * 5
        h1 = hp.set_ridge_params(5)
        cs, best, pred = eng.l1_qt(np.tile(L * (1 - h1) / h1, (1, 1)), [5, 5])
        locos.append((loco_from_predictions(pred[0], [1, 2]), int(best[0])))
        eng.close()
    assert locos[0][1] == locos[1][1]
    assert np.max(np.abs(locos[0][0] + locos[1][0])) < 1e-9 * np.max(np.abs(locos[0][0]))
