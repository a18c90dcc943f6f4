"""BGEN genotype blocks inflated and walked on the GPU (include/rg_bgen.h device path, csrc/bgen_inflate.hip) against zlib and numpy.

The arbiter of the inflate kernel is zlib itself (python's zlib module = the library the reference calls, Geno.cpp:2200-2210): every inflated byte
of every stream must be zlib's.  The walk is held to the bytes: integer dosages q = b1 + 2 b0 (units of 1 / 255), the exact integer sums, their
per-trait parts, and -- through them -- the doubles the host route accumulates in sample order (rg_bgen_read_dosages_info), to 1e-12."""
import os
import zlib

import numpy as np
import pytest

from oracle import bgen as obg
from regenie_amd.bgen import BgenDevice, BgenFile

pytestmark = pytest.mark.gpu


def _expect(blocks, n_file, file_idx, ref_first, mask=None):
    """numpy restatement of the walk over inflated blocks [nvar, 10 + 3 N]."""
    fi = np.arange(n_file) if file_idx is None else np.asarray(file_idx)
    miss = (blocks[:, 8:8 + n_file] & 0x80) != 0
    pr = blocks[:, 10 + n_file:10 + 3 * n_file].reshape(blocks.shape[0], n_file, 2).astype(np.int64)
    b0, b1 = pr[:, fi, 0], pr[:, fi, 1]
    ms = miss[:, fi]
    bx = np.where(b0 + b1 < 255, 255 - b0 - b1, 0) if ref_first else b0
    q = b1 + 2 * bx
    inf = 255 * (4 * bx + b1) - q * q
    obs = ~ms
    out = {"g16": np.where(ms, 0xFFFF, q).astype(np.uint16), "sum_q": (q * obs).sum(axis=1), "sum_info": (inf * obs).sum(axis=1),
           "n_obs": obs.sum(axis=1), "max_q": np.where(obs, q, 0).max(axis=1)}
    if mask is not None:
        mm = (np.asarray(mask) == 0)                                  # [P, n]: missing for the trait
        out["sum_q_t"] = np.stack([((q * obs) * mm[p]).sum(axis=1) for p in range(mm.shape[0])], axis=1)
        out["sum_info_t"] = np.stack([((inf * obs) * mm[p]).sum(axis=1) for p in range(mm.shape[0])], axis=1)
        out["n_obs_t"] = np.stack([(obs * mm[p]).sum(axis=1) for p in range(mm.shape[0])], axis=1)
    return out


def _check(res, want):
    assert (res["status"] == 0).all(), res["status"]
    for k, v in want.items():
        assert np.array_equal(res[k], v), k
    assert (res["g16_pad"] == 0).all()


@pytest.mark.parametrize("name", ["example.bgen", "example_3chr.bgen"])
def test_reference_fixture_streams(example_dir, name):
    """Every variant of the reference's own zlib fixtures: inflated bytes = zlib's, dosage rows and sums = the bytes'; the sums reproduce the
    doubles of the host route (the reference's arithmetic in sample order) to rounding."""
    with BgenFile(os.path.join(example_dir, name), threads=4) as f, BgenDevice(0) as d:
        m, n = f.n_variants, f.n_samples
        idx = np.arange(m)
        comp, off, clen, ulen = f.read_compressed(idx, threads=4)
        blocks = f.read_blocks(idx)
        d.set_samples(n)
        res = d.decode(comp, off, clen, ulen, fetch_raw=True)
        assert (res["raw"][:, :blocks.shape[1]] == blocks).all()
        for k in (0, m // 2, m - 1):
            assert zlib.decompress(comp[off[k]:off[k] + clen[k]].tobytes()) == res["raw"][k, :ulen[k]].tobytes()
        _check(res, _expect(blocks, n, None, False))
        # a subset of the samples in another order, ref-first, per-trait masks
        rng = np.random.default_rng(3)
        fi = rng.permutation(n)[: n - 37]
        mask = (rng.random((3, fi.size)) > 0.1).astype(np.uint8)
        d.set_samples(n, file_idx=fi, mask=mask)
        res = d.decode(comp, off, clen, ulen, ref_first=True, slot=1)
        _check(res, _expect(blocks, n, fi, True, mask))
        # the host route's doubles (dosage and info term per sample, summed in sample order)
        d.set_samples(n)
        res = d.decode(comp, off, clen, ulen)
        rows = np.empty((m, n)); info = np.empty((m, n))
        assert f.lib.rg_bgen_read_dosages_info(f.h, m, idx.ctypes.data, 0, rows.ctypes.data, info.ctypes.data, n) == 0
        obs = rows != -3.0
        tot = np.array([np.add.accumulate(np.where(obs[j], rows[j], 0.0))[-1] for j in range(m)])
        inf = np.array([np.add.accumulate(np.where(obs[j], info[j], 0.0))[-1] for j in range(m)])
        assert np.abs(res["sum_q"] / 255.0 - tot).max() <= 1e-12 * np.abs(tot).max()
        assert np.abs(res["sum_info"] / 65025.0 - inf).max() <= 1e-12 * max(1.0, np.abs(inf).max())


def _write(path, probs, miss, level, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=15):
    """BGEN v1.2 / layout 2 / 8 bits with zlib streams of a chosen level, strategy and window (oracle/bgen.py writes level 6 only)."""
    import struct
    m, n = miss.shape
    with open(path, "wb") as fh:
        fh.write(struct.pack("<I", 20))
        fh.write(struct.pack("<IIII", 20, m, n, 0x6e656762))
        fh.write(struct.pack("<I", 1 | (2 << 2)))
        for j in range(m):
            rs = ("rs%d" % j).encode()
            fh.write(struct.pack("<H", 0) + struct.pack("<H", len(rs)) + rs + struct.pack("<H", 1) + b"1" + struct.pack("<IH", 100 + j, 2))
            fh.write(struct.pack("<I", 1) + b"A" + struct.pack("<I", 1) + b"G")
            blk = struct.pack("<IHBB", n, 2, 2, 2) + np.where(miss[j], 0x82, 0x02).astype(np.uint8).tobytes() + bytes([0, 8]) + probs[j].tobytes()
            co = zlib.compressobj(level, zlib.DEFLATED, wbits, 8, strategy)
            z = co.compress(blk) + co.flush()
            fh.write(struct.pack("<II", len(z) + 4, len(blk)) + z)


@pytest.mark.parametrize("level", [1, 6, 9, 0])
def test_streams_of_every_kind(tmp_path, level):
    """60,000 samples (blocks of 180 KB: dozens of deflate blocks per stream, matches further back than the 8 KB the kernel keeps in LDS),
    hard calls with a share of genuine probabilities and missing samples, long runs (a monomorphic variant: matches of length 258 at distance 2),
    random bytes (literals only, long codes) -- at zlib levels 1 (what tools/bgen_e2e.py writes), 6, 9 and 0 (stored blocks)."""
    rng = np.random.default_rng(10 + level)
    m, n = 12, 60000
    maf = rng.uniform(0.02, 0.5, m)
    g = rng.binomial(2, maf[:, None], (m, n))
    p0 = np.where(g == 2, 255, 0); p1 = np.where(g == 1, 255, 0)
    soft = rng.random((m, n)) < 0.3
    a = rng.integers(0, 256, (m, n)); b = np.minimum(rng.integers(0, 256, (m, n)), 255 - a)
    p0 = np.where(soft, a, p0); p1 = np.where(soft, b, p1)
    p0[0], p1[0] = 255, 0                                             # monomorphic
    p0[1] = rng.integers(0, 256, n); p1[1] = rng.integers(0, 256, n)  # random bytes (prob0 + prob1 may exceed 1: max_q > 510 reports it)
    # a 24 KB stretch repeated: matches at distance ~24,000
    p0[2, 30000:54000], p1[2, 30000:54000] = p0[2, 3000:27000], p1[2, 3000:27000]
    probs = np.stack([p0, p1], axis=-1).astype(np.uint8)
    miss = rng.random((m, n)) < 0.002
    path = str(tmp_path / "s.bgen")
    _write(path, probs, miss, level)
    with BgenFile(path, threads=4) as f, BgenDevice(0) as d:
        idx = np.arange(m)
        comp, off, clen, ulen = f.read_compressed(idx)
        blocks = np.stack([np.frombuffer(zlib.decompress(comp[off[k]:off[k] + clen[k]].tobytes()), dtype=np.uint8) for k in range(m)])
        d.set_samples(n)
        res = d.decode(comp, off, clen, ulen, fetch_raw=True)
        assert (res["status"] == 0).all(), res["status"]
        assert (res["raw"][:, :blocks.shape[1]] == blocks).all()
        _check(res, _expect(blocks, n, None, False))
        assert res["max_q"][1] > 510 and res["max_q"][0] == 510


@pytest.mark.parametrize("strategy,wbits", [(zlib.Z_FIXED, 15), (zlib.Z_RLE, 15), (zlib.Z_HUFFMAN_ONLY, 15), (zlib.Z_FILTERED, 15),
                                            (zlib.Z_DEFAULT_STRATEGY, 9)])
def test_streams_of_other_writers(tmp_path, strategy, wbits):
    """What a writer other than regenie's test data may hold: blocks on the FIXED code (no tables in the stream), run-length matches only
    (distance 1, overlapping copies), literals only, the filtered strategy, and a 512-byte window (the zlib header's CINFO is not 7)."""
    rng = np.random.default_rng(77 + strategy + wbits)
    m, n = 6, 20000
    g = rng.binomial(2, rng.uniform(0.02, 0.5, m)[:, None], (m, n))
    p0 = np.where(g == 2, 255, 0); p1 = np.where(g == 1, 255, 0)
    soft = rng.random((m, n)) < 0.2
    a = rng.integers(0, 256, (m, n)); b = np.minimum(rng.integers(0, 256, (m, n)), 255 - a)
    p0 = np.where(soft, a, p0); p1 = np.where(soft, b, p1)
    p0[0], p1[0] = 0, 255                                             # all heterozygous: one long run
    probs = np.stack([p0, p1], axis=-1).astype(np.uint8)
    miss = rng.random((m, n)) < 0.01
    path = str(tmp_path / "o.bgen")
    _write(path, probs, miss, 6, strategy, wbits)
    with BgenFile(path, threads=2) as f, BgenDevice(0) as d:
        idx = np.arange(m)
        comp, off, clen, ulen = f.read_compressed(idx)
        blocks = np.stack([np.frombuffer(zlib.decompress(comp[off[k]:off[k] + clen[k]].tobytes()), dtype=np.uint8) for k in range(m)])
        d.set_samples(n)
        res = d.decode(comp, off, clen, ulen, fetch_raw=True)
        assert (res["status"] == 0).all(), res["status"]
        assert (res["raw"][:, :blocks.shape[1]] == blocks).all()
        _check(res, _expect(blocks, n, None, False))


def test_damaged_streams_are_flagged_not_decoded(tmp_path, example_dir):
    """A flipped bit inside a stream, a truncated stream, a wrong inflated length: the variant's status is non-zero (the caller then takes the
    host route, whose messages are the reference's) and the other variants of the batch are unaffected."""
    with BgenFile(os.path.join(example_dir, "example.bgen")) as f, BgenDevice(0) as d:
        idx = np.arange(8)
        comp, off, clen, ulen = f.read_compressed(idx)
        blocks = f.read_blocks(idx)
        d.set_samples(f.n_samples)
        c2 = comp.copy()
        c2[off[1] + clen[1] // 2] ^= 0x10          # corrupt data: a bad code, a bad distance or, at the latest, the checksum
        c2[off[3] + clen[3] - 2] ^= 0x01           # the Adler-32 trailer itself
        cl = clen.copy(); cl[5] -= 9               # truncated
        ul = ulen.copy(); ul[6] += 1               # the stream ends before the stated size
        res = d.decode(c2, off, cl, ul, fetch_raw=True)
        bad = {1, 3, 5, 6}
        for k in range(8):
            assert (res["status"][k] != 0) == (k in bad), (k, res["status"])
            if k not in bad:
                assert (res["raw"][k, :blocks.shape[1]] == blocks[k]).all()
