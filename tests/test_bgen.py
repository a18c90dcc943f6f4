"""BGEN v1.2 input (layout 2, 8-bit probabilities: the files regenie reads through its own fast path, Geno.cpp:1574-1699):
oracle and product reader against the reference's fixture pairs and against synthetic files.  Host-only, no GPU."""
import os

import numpy as np
import pytest

from oracle import bgen as obg
from regenie_amd.bgen import BgenFile
from regenie_amd.engine import RgError


def _bed_genotypes(prefix, m, n):
    raw = np.fromfile(prefix + ".bed", dtype=np.uint8)[3:].reshape(m, (n + 3) // 4)
    codes = ((raw[:, :, None] >> np.array([0, 2, 4, 6])) & 3).reshape(m, -1)[:, :n]
    return np.array([2.0, -3.0, 1.0, 0.0])[codes]


@pytest.mark.parametrize("bgen,bed,m", [("example.bgen", "example", 1000), ("example_3chr.bgen", "example_3chr", 500),
                                        ("example_3chr_zstd.bgen", "example_3chr", 500)])
def test_reference_fixture_pairs(example_dir, bgen, bed, m):
    """The reference ships the same genotypes as .bed and .bgen (zlib and zstd); its tests compare runs on both
    (test/test_bash.sh:143-216).  Every dosage must be the .bed genotype, for the oracle and for the product reader."""
    want = _bed_genotypes(os.path.join(example_dir, bed), m, 500)
    o = obg.BgenOracle(os.path.join(example_dir, bgen))
    assert (o.m, o.n, o.layout) == (m, 500, 2)
    got = np.stack([o.dosages(j) for j in range(m)])
    assert (got == want).all()
    with BgenFile(os.path.join(example_dir, bgen), threads=3) as f:
        assert (f.n_variants, f.n_samples) == (m, 500) and f.compression == o.compression
        rows = f.read_dosages(np.arange(m))
        assert (rows == want).all()
        assert (f.read_dosages([m - 1, 0, 7]) == want[[m - 1, 0, 7]]).all()
        assert (f.read_dosages([3], ref_first=True) == np.where(want[3] == -3, -3, 2 - want[3])).all()
        ids = f.sample_ids()
        assert ids == o.sample_ids and (not ids or ids[:2] == ["1_1", "2_2"])
        bim = [ln.split() for ln in open(os.path.join(example_dir, bed + ".bim")).read().split("\n") if ln]
        for j in (0, 1, m // 2, m - 1):
            v = f.variant(j)
            assert v["rsid"] == bim[j][1] and v["chrom"] == bim[j][0] and v["pos"] == int(bim[j][3])
            assert v["offset"] == o.variants[j]["offset"]


def test_synthetic_probabilities_and_missing(tmp_path):
    rng = np.random.default_rng(2)
    m, n = 70, 333
    p0 = rng.integers(0, 256, (m, n))
    p1 = np.minimum(rng.integers(0, 256, (m, n)), 255 - p0)
    probs = np.stack([p0, p1], axis=-1).astype(np.uint8)
    miss = rng.random((m, n)) < 0.03
    variants = [(1 + j // 40, 100 + j, "rs%d" % j, "A", "G") for j in range(m)]
    a, b, d = p0 / 255.0, p1 / 255.0, None
    c = np.maximum(1 - a - b, 0)
    want = np.where(miss, -3.0, b + 2 * a)
    want_rf = np.where(miss, -3.0, b + 2 * c)
    for comp, ids in ((1, ["%d_%d" % (i, i) for i in range(n)]), (0, None)):
        path = str(tmp_path / ("s%d.bgen" % comp))
        obg.write_bgen(path, probs, miss, variants, sample_ids=ids, compression=comp)
        o = obg.BgenOracle(path)
        assert all((o.dosages(j) == want[j]).all() for j in range(m))
        with BgenFile(path, threads=4) as f:
            assert f.has_sample_ids == (ids is not None) and f.compression == comp
            assert (f.read_dosages(np.arange(m)) == want).all()
            assert (f.read_dosages(np.arange(m), ref_first=True) == want_rf).all()
            assert f.variant(41) == dict(chrom="2", pos=141, rsid="rs41", a0="A", a1="G", offset=o.variants[41]["offset"])
            blk = f.read_blocks([5, 0, m - 1])                    # rg_bgen_read_blocks: the bytes the Step-2 driver walks itself
            assert blk.shape == (3, 10 + 3 * n)
            for k, j in enumerate((5, 0, m - 1)):
                assert blk[k, :4].view("<u4")[0] == n and tuple(blk[k, 4:8]) == (2, 0, 2, 2) and tuple(blk[k, 8 + n:10 + n]) == (0, 8)
                assert ((blk[k, 8:8 + n] & 0x80 != 0) == miss[j]).all() and ((blk[k, 8:8 + n] & 0x3f) == 2).all()
                assert (blk[k, 10 + n:].reshape(n, 2)[~miss[j]] == probs[j][~miss[j]]).all()


def test_refusals_and_damage(tmp_path, example_dir):
    def err(path):
        with pytest.raises(RgError) as e:
            BgenFile(path)
        return e.value
    assert "magic" in str(err(os.path.join(example_dir, "example.bed"))) or err(os.path.join(example_dir, "example.bed")).code == -2
    assert "cannot open" in str(err(str(tmp_path / "nope.bgen")))
    raw = bytearray(open(os.path.join(example_dir, "example_3chr.bgen"), "rb").read())
    # layout 1 flag
    lay1 = bytearray(raw)
    lay1[20] = (lay1[20] & ~0x3C) | (1 << 2)
    p = str(tmp_path / "l1.bgen")
    open(p, "wb").write(lay1)
    e = err(p)
    assert e.code == -3 and "layout 1 is not supported" in str(e)
    # truncated file
    p = str(tmp_path / "t.bgen")
    open(p, "wb").write(raw[:len(raw) - 100])
    assert err(p).code == -2
    # a damaged variant count (bytes 8 - 11 of the header) is refused before the variant table is reserved for it
    big = bytearray(raw)
    big[11] = 0x70
    open(p, "wb").write(big)
    e = err(p)
    assert e.code == -2 and "more variants than the file can hold" in str(e)
    # ... and a block whose stated inflated length no stream of its stored length can reach, before the buffer for it exists
    o0 = obg.BgenOracle(os.path.join(example_dir, "example_3chr.bgen"))
    far = bytearray(raw)
    at0 = o0.variants[2]["data"]
    clen = int.from_bytes(raw[at0:at0 + 4], "little")
    far[at0 + 4:at0 + 8] = (1100 * clen + 65).to_bytes(4, "little")
    open(p, "wb").write(far)
    with BgenFile(p) as f:
        with pytest.raises(RgError) as e2:
            f.read_dosages([2])
        assert e2.value.code == -2
    # a damaged compressed block: the reference's message (Geno.cpp:1616-1617)
    o = obg.BgenOracle(os.path.join(example_dir, "example_3chr.bgen"))
    bad = bytearray(raw)
    at = o.variants[5]["data"] + 8
    bad[at + 2: at + 12] = bytes(10)
    p = str(tmp_path / "z.bgen")
    open(p, "wb").write(bad)
    with BgenFile(p) as f:
        with pytest.raises(RgError, match="failed to decompress genotype data block for variant: " + o.variants[5]["rsid"]):
            f.read_dosages([5])
        assert f.read_dosages([4]).shape == (1, 500)
        with pytest.raises(RgError):
            f.read_dosages([500])
        # the block form (rg_bgen_read_blocks: own DEFLATE decoder first, zlib as the arbiter) refuses the same block with the same message,
        # serves its neighbours, and from several caller threads at once every caller sees its own error
        with pytest.raises(RgError, match="failed to decompress genotype data block for variant: " + o.variants[5]["rsid"]):
            f.read_blocks([4, 5, 6])
        good = f.read_blocks([4, 6])
        assert good.shape == (2, 10 + 3 * 500) and (f.read_blocks([6])[0] == good[1]).all()
        with pytest.raises(RgError):
            f.read_blocks([500])
        import threading
        seen = {}

        def worker(k):
            try:
                seen[k] = f.read_blocks([5] if k % 2 else [4, 6]).shape
            except RgError as ex:
                seen[k] = str(ex)
        th = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert all(seen[k] == (2, 1510) for k in range(0, 8, 2))
        assert all("failed to decompress" in seen[k] and o.variants[5]["rsid"] in seen[k] for k in range(1, 8, 2))
    # every block of the three fixture files through both decoders: RG_BGEN_ZLIB=1 is read at the first inflate of a process, so the
    # comparison is against the dosage rows, which the oracle pins (test_reference_fixture_pairs)
    for name in ("example.bgen", "example_3chr.bgen"):
        with BgenFile(os.path.join(example_dir, name), threads=2) as f:
            idx = np.arange(f.n_variants)
            blk = f.read_blocks(idx)
            n = f.n_samples
            pr = blk[:, 10 + n:].reshape(len(idx), n, 2).astype(np.float64) / 255.0
            dos = np.where(blk[:, 8:8 + n] & 0x80, -3.0, pr[:, :, 1] + 2.0 * pr[:, :, 0])
            assert (dos == f.read_dosages(idx)).all()


# ---- the host driver's --bgen / --sample handling (runs before any device is touched) ---------------------------------------
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "regenie_amd", "bin", "regenie-amd")


def _cli(args, cwd):
    import subprocess
    from regenie_amd import build
    build.build()
    return subprocess.run([BIN, "--step", "1", "--bsize", "100", "--out", os.path.join(cwd, "o")] + args, cwd=cwd, capture_output=True,
                          text=True, timeout=120)


def test_cli_bgen_input_messages(example_dir, tmp_path):
    """Messages of prep_bgen / read_bgen_sample (Geno.cpp:38-175, :395-456)."""
    E = example_dir
    ph = ["--phenoFile", os.path.join(E, "phenotype.txt")]
    cwd = str(tmp_path)
    r = _cli(["--bgen", os.path.join(E, "example.bgen")] + ph, cwd)
    assert " * bgen" in r.stdout and "with 500 named samples and 1000 variants with 8-bit encoding" in r.stdout
    assert "-n_snps = 1000" in r.stdout and "-n_samples = 500" in r.stdout
    sample = open(os.path.join(E, "example_3chr.sample")).read().split("\n")

    def with_sample(lines):
        p = str(tmp_path / "s.sample")
        open(p, "w").write("\n".join(lines))
        r = _cli(["--bgen", os.path.join(E, "example_3chr.bgen"), "--sample", p] + ph, cwd)
        assert r.returncode != 0
        return r.stdout

    assert "ERROR: number of samples in BGEN file does not match that in the sample file." in with_sample(sample[:-3] + [""])
    assert "ERROR: header of the sample file must start with: ID_1 ID_2" in with_sample(["FID IID missing"] + sample[1:])
    assert "ERROR: second line of sample file must start with: 0 0." in with_sample(sample[:1] + ["1 1 0 D P"] + sample[2:])
    assert "ERROR: duplicate individual in bgen file : FID_IID =1_1" in with_sample(sample[:3] + [sample[2]] + sample[4:])
    # a file without embedded identifiers needs --sample; an unknown chromosome code is refused
    rng = np.random.default_rng(0)
    probs = np.zeros((5, 8, 2), np.uint8)
    probs[:, :, 0] = 255
    variants = [("1", 10 + j, "v%d" % j, "A", "C") for j in range(5)]
    p = str(tmp_path / "noids.bgen")
    obg.write_bgen(p, probs, np.zeros((5, 8), bool), variants, sample_ids=None)
    r = _cli(["--bgen", p] + ph, cwd)
    assert r.returncode != 0 and "ERROR: bgen file has no sample identifiers; specify a sample file with --sample" in r.stdout
    variants[2] = ("chrUn", 12, "v2", "A", "C")
    p = str(tmp_path / "badchr.bgen")
    obg.write_bgen(p, probs, np.zeros((5, 8), bool), variants, sample_ids=["%d_%d" % (i, i) for i in range(8)])
    r = _cli(["--bgen", p] + ph, cwd)
    assert r.returncode != 0 and "ERROR: unknown chromosome code in bgen file." in r.stdout
    # more than one genotype input
    r = _cli(["--bgen", os.path.join(E, "example.bgen"), "--bed", os.path.join(E, "example")] + ph, cwd)
    assert r.returncode != 0 and "ERROR: must use either --bed,--bgen or --pgen." in r.stdout


def test_zlib_arbiter_path_gives_the_same_rows(example_dir):
    """RG_BGEN_ZLIB=1 sends every block through zlib (the arbiter of whatever csrc/inflate_fast.h does not accept); the variable is read
    once per process, so the comparison runs in a child: same dosage rows and same probability blocks as the default decoder."""
    import subprocess
    import sys
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from regenie_amd.bgen import BgenFile\n"
            "f = BgenFile(%r, threads=2); r = f.read_dosages(np.arange(f.n_variants)); b = f.read_blocks(np.arange(0, f.n_variants, 7))\n"
            "import hashlib; print(hashlib.sha256(r.tobytes()).hexdigest(), hashlib.sha256(b.tobytes()).hexdigest())\n"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(example_dir, "example.bgen")))
    outs = []
    for env in ({}, {"RG_BGEN_ZLIB": "1"}):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env={**os.environ, **env})
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.split())
    assert outs[0] == outs[1] and len(outs[0]) == 2


def test_short_lived_reader_threads_do_not_leak(example_dir):
    """The read calls start their worker threads per call (the Step-2 driver does so per block); the decoder tables a worker allocates die with
    it.  A leak of the 44 KB tables per thread showed as ~90 MB over these 2,000 threads; 20 MB of growth is allowed for allocator noise."""
    import psutil
    with BgenFile(os.path.join(example_dir, "example.bgen"), threads=8) as f:
        idx = np.arange(64)
        f.read_blocks(idx)
        rss0 = psutil.Process().memory_info().rss
        for _ in range(250):
            f.read_blocks(idx)
        grown = psutil.Process().memory_info().rss - rss0
    assert grown < 20 << 20, grown


def test_read_compressed_hands_out_the_stored_streams(example_dir):
    """rg_bgen_read_compressed (what the device decoder is fed): every stream sits at a 16-byte aligned offset and zlib inflates it to the block
    rg_bgen_read_blocks returns; a zstd file is refused (the device decoder takes zlib)."""
    import zlib
    with BgenFile(os.path.join(example_dir, "example.bgen"), threads=3) as f:
        idx = np.array([0, 1, 2, 500, 999, 7])
        buf, off, clen, ulen = f.read_compressed(idx, threads=3)
        blk = f.read_blocks(idx)
        assert (off % 16 == 0).all() and (ulen == blk.shape[1]).all()
        for k in range(idx.size):
            assert zlib.decompress(buf[off[k]:off[k] + clen[k]].tobytes()) == blk[k].tobytes()
    with BgenFile(os.path.join(example_dir, "example_3chr_zstd.bgen")) as f:
        with pytest.raises(Exception) as e:
            f.read_compressed([0])
        assert "zlib" in str(e.value)
