"""PLINK2 .pgen hardcall input: oracle, product reader and -- when it has been
built from /root/reference by oracle/Makefile -- the reference's own pgenlib, on the reference's fixture pair
and on synthetic files that exercise every record type.  Host-only: runs without a GPU."""
import ctypes
import os

import time

import numpy as np
import pytest

from oracle import pgen as opg
from regenie_amd.engine import RgError
from regenie_amd.pgen import PgenFile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libpgen_ref.so")


def _bed_rows(prefix, m, n):
    raw = np.fromfile(prefix + ".bed", dtype=np.uint8)
    assert raw[:3].tolist() == [0x6C, 0x1B, 0x01]
    return raw[3:].reshape(m, (n + 3) // 4)


def _ref_hardcalls(path, n, m):
    lib = ctypes.CDLL(REF_LIB)
    idx = np.arange(m, dtype=np.int64)
    out = np.zeros((m, n))
    lib.pgen_ref_hardcalls(path.encode(), ctypes.c_uint32(n), None, ctypes.c_int64(0), idx.ctypes.data_as(ctypes.c_void_p),
                           ctypes.c_int64(m), out.ctypes.data_as(ctypes.c_void_p))
    return out


def _ref_counts(path, n):
    lib = ctypes.CDLL(REF_LIB)
    c = (ctypes.c_int64 * 4)()
    lib.pgen_ref_counts(path.encode(), ctypes.c_uint32(n), c)
    return list(c)


needs_ref = pytest.mark.skipif(not os.path.exists(REF_LIB), reason="oracle/_ref/libpgen_ref.so not built (needs /root/reference)")


def synth(m, n, seed):
    """Genotypes (pgen codes) and a record type per variant, every type 0..7 present when n allows difflists."""
    rng = np.random.default_rng(seed)
    lim = n // opg.MAX_DIFFLIST_DIV
    g = np.zeros((m, n), dtype=np.uint8)
    vts = []
    prev = None
    cycle = [0, 1, 2, 3, 4, 2, 2, 3, 6, 7, 5, 1, 3, 0, 2]
    for j in range(m):
        t = cycle[j % len(cycle)]
        if lim == 0 and t in (2, 3, 4, 6, 7):
            t = 0 if j % 2 else 1
        if t in (2, 3) and prev is None:
            t = 0
        k = int(rng.integers(0, lim + 1)) if lim else 0
        pos = rng.choice(n, size=k, replace=False)
        if t == 0:
            maf = rng.uniform(0.05, 0.5)
            row = ((rng.random(n) < maf).astype(np.uint8) + (rng.random(n) < maf).astype(np.uint8))
            row[rng.random(n) < 0.02] = 3
        elif t == 1:
            a, b = sorted(rng.choice(4, size=2, replace=False).tolist())
            row = np.where(rng.random(n) < 0.4, b, a).astype(np.uint8)
            row[pos] = rng.integers(0, 4, k)
        elif t == 2:
            row = prev.copy()
            row[pos] = rng.integers(0, 4, k)
        elif t == 3:
            row = opg._invert(prev)
            row[pos] = rng.integers(0, 4, k)
            if (opg._invert(row) != prev).sum() > lim:
                row = opg._invert(prev)
        elif t == 5:
            row = np.zeros(n, dtype=np.uint8)
        else:
            row = np.full(n, t & 3, dtype=np.uint8)
            row[pos] = rng.integers(0, 4, k)
        if t == 1:  # the exceptions of a one-bit record are the samples outside its two commonest codes
            cnt = np.bincount(row, minlength=4)
            if n - np.sort(cnt)[-2:].sum() > lim:
                t = 0
        g[j] = row
        vts.append(t)
        if (t & 6) != 2:
            prev = row
    return g, vts


def test_oracle_on_reference_fixture_pair(example_dir):
    """example.pgen and example.bed hold the same genotypes (test/test_bash.sh:411-433 relies on it)."""
    o = opg.PgenOracle(os.path.join(example_dir, "example.pgen"))
    assert (o.m, o.n, o.max_alleles, o.dosage_present) == (1000, 500, 2, False)
    bed = _bed_rows(os.path.join(example_dir, "example"), 1000, 500)
    assert all((o.bed_row(j) == bed[j]).all() for j in range(o.m))


def test_product_on_reference_fixture_pair(example_dir):
    with PgenFile(os.path.join(example_dir, "example.pgen")) as f:
        assert (f.n_variants, f.n_samples, f.max_alleles, f.phase_present) == (1000, 500, 2, False)
        bed = _bed_rows(os.path.join(example_dir, "example"), 1000, 500)
        assert (f.read_bed_rows(np.arange(1000)) == bed).all()
        idx = np.array([999, 0, 500, 500, 3])
        assert (f.read_bed_rows(idx) == bed[idx]).all()
        hc = f.read_hardcalls(7)
        assert (hc == np.array([2.0, -3.0, 1.0, 0.0])[(bed[7][:, None] >> np.array([0, 2, 4, 6])) & 3].reshape(-1)[:500]).all()


@needs_ref
def test_reference_reader_on_fixture_pair(example_dir):
    p = os.path.join(example_dir, "example.pgen")
    assert _ref_counts(p, 500) == [500, 1000, 2, 0]
    o = opg.PgenOracle(p)
    ref = _ref_hardcalls(p, 500, 1000)
    assert all((o.hardcalls(j) == ref[j]).all() for j in range(1000))


CASES = [  # m, n, reclen bytes, 8-bit vrtypes, phase track, nonref storage, mode
    (60, 8, 1, False, False, 0, 0x10),
    (300, 777, 2, False, False, 1, 0x10),      # odd bytes per row
    (300, 1001, 2, True, True, 3, 0x10),       # phase tracks to step over, stored nonref flags
    (240, 4096, 3, True, False, 2, 0x11),
    (90, 70001, 4, False, False, 0, 0x10),     # 3-byte sample ids in difflists
    (66000, 130, 1, False, False, 3, 0x10),    # two header blocks
]


@pytest.mark.parametrize("m,n,rl,wide,phase,nonref,mode", CASES)
def test_every_record_type(tmp_path, m, n, rl, wide, phase, nonref, mode):
    g, vts = synth(m, n, seed=m + n)
    if n >= 64:
        assert set(vts) == set(range(8))
    path = str(tmp_path / "s.pgen")
    opg.write_pgen(path, g, vts, reclen_bytes=rl, wide_vrtypes=wide, phase=phase, nonref=nonref, mode=mode, seed=3)
    truth = opg.HARDCALL[g]
    want_rows = np.stack([np.frombuffer(opg._pack2(opg.PGEN_TO_BED[g[j]]), np.uint8) for j in range(m)]) if m <= 400 else None
    step = 1 if m <= 400 else 97
    o = opg.PgenOracle(path)
    assert (o.m, o.n, o.phase_present) == (m, n, phase)
    for j in range(0, m, step):
        assert (o.hardcalls(j) == truth[j]).all(), (j, vts[j])
    with PgenFile(path) as f:
        assert (f.n_variants, f.n_samples, f.phase_present) == (m, n, phase)
        rows = f.read_bed_rows(np.arange(m))
        codes = (rows[:, :, None] >> np.array([0, 2, 4, 6])) & 3
        assert (np.array([2.0, -3.0, 1.0, 0.0])[codes].reshape(m, -1)[:, :n] == truth).all()
        if n & 3:
            assert ((rows[:, -1] >> (2 * (n & 3))) == 0).all()  # padding bits zero, like a .bed written by plink
        if want_rows is not None:
            assert (rows == want_rows).all()
        # any order: LD-compressed variants find their base without having been read in sequence
        perm = np.random.default_rng(5).permutation(m)[:300]
        assert (f.read_bed_rows(perm) == rows[perm]).all()
        assert (f.read_hardcalls(int(perm[0])) == truth[perm[0]]).all()
        # worker threads (the reference decodes a block's variants under OpenMP): same rows, each worker with its own LD cache
        f.set_threads(5)
        assert (f.read_bed_rows(np.arange(m)) == rows).all()
        assert (f.read_bed_rows(perm) == rows[perm]).all()
    if os.path.exists(REF_LIB):  # the writer is format-conformant: regenie's own reader gets the genotypes back
        assert _ref_counts(path, n) == [n, m, 2, 0]
        assert (_ref_hardcalls(path, n, m) == truth).all()


def _ref_dosages(path, n, m):
    lib = ctypes.CDLL(REF_LIB)
    idx = np.arange(m, dtype=np.int64)
    out = np.zeros((m, n))
    lib.pgen_ref_dosages(path.encode(), ctypes.c_uint32(n), idx.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(m),
                         out.ctypes.data_as(ctypes.c_void_p))
    return out


@pytest.mark.parametrize("m,n,phase", [(120, 500, False), (121, 1003, True), (122, 261, "explicit"), (40, 70001, "explicit")])
def test_dosage_tracks_read_like_the_reference(tmp_path, m, n, phase):
    """PgenReader::Read -- what regenie calls once a file has dosages (Geno.cpp:1101, :1795-1796): the three dosage layouts,
    behind every main-track record type and behind both forms of the phase track."""
    g, vts = synth(m, n, seed=m + n)
    rng = np.random.default_rng(m)
    dos = {}
    truth = opg.HARDCALL[g].copy()
    for j in range(m):
        kind = [None, 0x20, 0x40, 0x60][j % 4]
        if kind is None:
            continue
        k = int(rng.integers(0, (n // 8 if kind == 0x20 else n) + 1))
        if j in (1, 2, 3):
            k = 0                                               # empty list / all-65535 / all-zero bit array
        ids = np.sort(rng.choice(n, size=k, replace=False))
        vals = rng.integers(0, 32769, ids.size).astype(np.uint16)
        dos[j] = (kind, ids, vals)
        truth[j, ids] = vals / 16384.0
    path = str(tmp_path / "d.pgen")
    opg.write_pgen(path, g, vts, wide_vrtypes=True, phase=phase, dosage=dos, reclen_bytes=3, seed=m)
    o = opg.PgenOracle(path)
    assert o.dosage_present and o.phase_present == bool(phase)
    with PgenFile(path) as f:
        assert f.dosage_present and f.n_variants == m
        order = list(range(m)) + [int(x) for x in rng.permutation(m)[:50]]   # in sequence, then out of order (LD bases)
        for j in order:
            assert (f.read_dosages(j) == truth[j]).all(), (j, vts[j], dos.get(j, (0,))[0])
        f.set_threads(4)
        assert (f.read_dosage_rows(np.arange(m)) == truth).all()                # a block at once, over worker threads
        pick = rng.permutation(m)[:40]
        assert (f.read_dosage_rows(pick) == truth[pick]).all()
        for j in range(0, m, 3):
            assert (o.dosages(j) == truth[j]).all()
            assert (f.read_hardcalls(j) == opg.HARDCALL[g[j]]).all()        # ReadHardcalls still ignores the dosages
        with pytest.raises(RgError) as e:                                   # 2-bit rows are not what regenie would use here
            f.read_bed_rows([0])
        assert e.value.code == -3 and "dosages" in str(e.value)
    if os.path.exists(REF_LIB):
        assert _ref_counts(path, n) == [n, m, 2, 1]
        assert (_ref_dosages(path, n, m) == truth).all()
    # a record cut short inside its dosage values is an error that names the variant
    j = max(k for k in dos if dos[k][0] == 0x60 and dos[k][1].size)
    raw = bytearray(open(path, "rb").read())
    end = int(o.fpos[j + 1])
    lens_at = 12 + 8 + m                                         # 8-bit vrtypes, then 3-byte record lengths
    ln = int.from_bytes(raw[lens_at + 3 * j: lens_at + 3 * j + 3], "little")
    raw[lens_at + 3 * j: lens_at + 3 * j + 3] = (ln - 2).to_bytes(3, "little")
    del raw[end - 2:end]
    bad = str(tmp_path / "cut.pgen")
    open(bad, "wb").write(raw)
    with PgenFile(bad) as f:
        with pytest.raises(RgError) as e:
            f.read_dosages(j)
        assert e.value.code == -2 and "variant %d" % (j + 1) in str(e.value)


def test_dosages_of_a_hardcall_file_are_its_hardcalls(example_dir):
    with PgenFile(os.path.join(example_dir, "example.pgen")) as f:
        assert not f.dosage_present
        for j in (0, 1, 500, 999):
            assert (f.read_dosages(j) == f.read_hardcalls(j)).all()


def test_fixed_width_mode(tmp_path):
    g, _ = synth(50, 333, seed=9)
    path = str(tmp_path / "f.pgen")
    opg.write_pgen_fixed(path, g)
    o = opg.PgenOracle(path)
    with PgenFile(path) as f:
        for j in range(50):
            assert (o.hardcalls(j) == opg.HARDCALL[g[j]]).all()
            assert (f.read_hardcalls(j) == opg.HARDCALL[g[j]]).all()
    if os.path.exists(REF_LIB):
        assert (_ref_hardcalls(path, 333, 50) == opg.HARDCALL[g]).all()


def _open_error(path):
    with pytest.raises(RgError) as e:
        PgenFile(path)
    return e.value


def test_refusals(tmp_path, example_dir):
    g, vts = synth(40, 200, seed=1)
    # dosage track: regenie would switch to dosages (Geno.cpp:1101), so a 2-bit reader must not accept the file
    p = str(tmp_path / "d.pgen")
    opg.write_pgen(p, g, vts, wide_vrtypes=True, dosage_variant=17)
    with PgenFile(p) as f:
        assert f.dosage_present
        with pytest.raises(RgError) as e:
            f.read_bed_rows([0])
        assert e.value.code == -3 and "dosages" in str(e.value)
    assert opg.PgenOracle(p).dosage_present
    if os.path.exists(REF_LIB):
        assert _ref_counts(p, 200)[3] == 1
    # allele-count bytes in the header (a file that may hold multiallelic variants): PgenReader::Load stops the
    # run on them even when every count is 2 (pgenlibr.cpp:65-68) -- so does this reader, with the same words.
    # (Not cross-checked against oracle/_ref here: the reference reports it with exit(-1).)
    for ac5 in (3, 2):
        p = str(tmp_path / ("m%d.pgen" % ac5))
        ac = np.full(40, 2)
        ac[5] = ac5
        opg.write_pgen(p, g, vts, allele_counts=ac)
        err = _open_error(p)
        assert err.code == -3 and "only bi-allelic variants should be present" in str(err)
        with pytest.raises(opg.PgenError):
            opg.PgenOracle(p)
    # not a pgen / a bed file / fixed-width dosage modes
    assert "magic" in str(_open_error(os.path.join(example_dir, "example.bim")))
    assert "--bed" in str(_open_error(os.path.join(example_dir, "example.bed")))
    p = str(tmp_path / "m3.pgen")
    open(p, "wb").write(b"\x6c\x1b\x03" + b"\x00" * 20)
    assert _open_error(p).code == -3                                    # fixed-width dosage storage modes are not read
    assert "cannot open" in str(_open_error(str(tmp_path / "nope.pgen")))


def test_malformed_files_are_errors_not_crashes(tmp_path):
    g, vts = synth(40, 200, seed=2)
    p = str(tmp_path / "ok.pgen")
    opg.write_pgen(p, g, vts)
    raw = bytearray(open(p, "rb").read())
    o = opg.PgenOracle(p)
    # truncated file: the header promises more bytes than there are
    t = str(tmp_path / "trunc.pgen")
    open(t, "wb").write(raw[:len(raw) - 50])
    assert _open_error(t).code == -2
    # a damaged variant count (one flipped byte: 40 -> 1.96e9 variants) is refused at once, not after 18 GB of per-variant tables were
    # allocated for it (found by flipping bytes of small files under AddressSanitizer: the open call took minutes)
    big = bytearray(raw)
    big[6] = 117
    open(t, "wb").write(big)
    t0 = time.time()
    e = _open_error(t)
    assert e.code == -2 and "more variants than bytes" in str(e) and time.time() - t0 < 2.0
    # a difflist whose sample index leaves the file's range
    j = vts.index(4)
    rec0 = int(o.fpos[j])
    bad = bytearray(raw)
    assert bad[rec0] >= 1          # difflist length, then the first group's first sample id (1 byte for n=200)
    bad[rec0 + 1] = 250
    b = str(tmp_path / "bad.pgen")
    open(b, "wb").write(bad)
    with PgenFile(b) as f:
        with pytest.raises(RgError) as e:
            f.read_bed_rows([j])
        assert e.value.code == -2 and "variant %d" % (j + 1) in str(e.value)
        assert (f.read_hardcalls(0) == opg.HARDCALL[g[0]]).all()  # the handle survives the error
        f.set_threads(3)
        with pytest.raises(RgError) as e:
            f.read_bed_rows(np.arange(40))
        assert "variant %d" % (j + 1) in str(e.value)
        with pytest.raises(RgError):
            f.set_threads(0)
    with pytest.raises(opg.PgenError):
        opg.PgenOracle(b).codes(j)
    # a difflist longer than N/8
    bad = bytearray(raw)
    bad[rec0] = 127
    open(b, "wb").write(bad)
    with PgenFile(b) as f:
        with pytest.raises(RgError):
            f.read_bed_rows([j])
    # index out of range
    with PgenFile(p) as f:
        with pytest.raises(RgError):
            f.read_bed_rows([40])
        with pytest.raises(RgError):
            f.read_hardcalls(-1)


# ---- the host driver's pvar / psam / pgen checks (they run before any device is touched) -------------------
BIN = os.path.join(ROOT, "regenie_amd", "bin", "regenie-amd")


def _cli(prefix, example_dir, cwd):
    import subprocess
    from regenie_amd import build
    build.build()
    return subprocess.run([BIN, "--step", "1", "--pgen", prefix, "--phenoFile", os.path.join(example_dir, "phenotype.txt"),
                           "--bsize", "100", "--out", os.path.join(cwd, "o")], cwd=cwd, capture_output=True, text=True, timeout=120)


def _copy_example(example_dir, tmp_path):
    import shutil
    for ext in ("pgen", "pvar", "psam"):
        shutil.copy(os.path.join(example_dir, "example." + ext), str(tmp_path / ("x." + ext)))
    return str(tmp_path / "x")


def test_cli_pgen_input_errors_like_reference(example_dir, tmp_path):
    """Messages of read_pvar / read_psam / prep_pgen (Geno.cpp:771-1103)."""
    pfx = _copy_example(example_dir, tmp_path)
    pvar = open(pfx + ".pvar").read().split("\n")
    psam = open(pfx + ".psam").read().split("\n")

    def run_with(pvar_lines=None, psam_lines=None):
        open(pfx + ".pvar", "w").write("\n".join(pvar_lines if pvar_lines is not None else pvar))
        open(pfx + ".psam", "w").write("\n".join(psam_lines if psam_lines is not None else psam))
        r = _cli(pfx, example_dir, str(tmp_path))
        assert r.returncode != 0
        return r.stdout

    assert "ERROR: number of variants in pgen file and pvar file don't match." in run_with(pvar_lines=pvar[:-3] + [""])
    assert "ERROR: number of samples in pgen file and psam file don't match." in run_with(psam_lines=psam[:-3] + [""])
    assert "ERROR: header of pvar file does not have correct format." in run_with(pvar_lines=["#CHROM\tPOS\tNAME\tREF\tALT"] + pvar[1:])
    assert "ERROR: incorrectly formatted pvar file at line 3" in run_with(pvar_lines=pvar[:3] + ["1\t3\t3"] + pvar[4:])
    assert "ERROR: unknown chromosome code in pvar file at line 2" in run_with(pvar_lines=pvar[:2] + ["chrUn\t2\t2\t2\t1"] + pvar[3:])
    assert "ERROR: chromosomes in pvar file are not in ascending order." in run_with(pvar_lines=pvar[:2] + ["2\t2\t2\t2\t1"] + pvar[3:])
    assert "ERROR: invalid header (must start with #FID [not #IID])." in run_with(psam_lines=["#IID\tSEX"] + psam[1:])
    assert "ERROR: header does not have the correct format." in run_with(psam_lines=["#FID\tID\tSEX"] + psam[1:])
    assert "ERROR: duplicate individual in fam file : FID_IID=1_1" in run_with(psam_lines=psam[:2] + [psam[1]] + psam[3:])
    assert "ERROR: unrecognized sex code in file : 'F'" in run_with(psam_lines=psam[:1] + ["1\t1\tF\t0.1"] + psam[2:])
    # leading "##" meta lines are skipped, a blank one is an error
    assert "ERROR: no blank lines should be before the header line in pvar file." in run_with(pvar_lines=["##fileformat=x", ""] + pvar)
    # a file with dosage tracks is run in dosage mode (Geno.cpp:1101): the driver announces the fp64 level-0 path
    # (end to end on a GPU: tests/test_cli_gpu.py::test_cli_pgen_dosages)
    g, vts = synth(30, 500, seed=3)
    dpfx = str(tmp_path / "dos")
    opg.write_pgen(dpfx + ".pgen", g, vts, wide_vrtypes=True, dosage={4: (0x40, np.arange(500), np.full(500, 8192, np.uint16))})
    opg.write_pvar_psam(dpfx, [1] * 30, 500)
    import subprocess as sp
    r = sp.run([BIN, "--step", "1", "--pgen", dpfx, "--phenoFile", os.path.join(example_dir, "phenotype.txt"), "--bsize", "10"],
               cwd=str(tmp_path), capture_output=True, text=True)
    assert "-dosages present: level 0 runs on the fp64 genotype path" in r.stdout
    # both inputs at once
    import subprocess
    r = subprocess.run([BIN, "--step", "1", "--pgen", pfx, "--bed", os.path.join(example_dir, "example"), "--phenoFile",
                        os.path.join(example_dir, "phenotype.txt"), "--bsize", "100"], cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode != 0 and "ERROR: must use either --bed,--bgen or --pgen." in r.stdout


def test_cli_pgen_reads_files_then_needs_a_gpu(example_dir, tmp_path):
    """With intact files the driver gets through pvar/psam/pgen (meta lines before the headers included); without a GPU
    it then stops at device creation -- there is no CPU path to fall back to."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_cli_gpu.py::test_cli_pgen_equals_bed")
    pfx = _copy_example(example_dir, tmp_path)
    pvar = open(pfx + ".pvar").read()
    open(pfx + ".pvar", "w").write("##fileformat=PVARv1.0\n##contig=<ID=1>\n" + pvar)
    r = _cli(pfx, example_dir, str(tmp_path))
    assert "n_snps = 1000" in r.stdout and "n_samples = 500" in r.stdout and " * pgen" in r.stdout
    assert r.returncode != 0 and "no MI355X / HIP device available" in r.stdout


# ---- property test: any genotype matrix, any admissible record type per variant, any header layout -------------------------
try:
    from hypothesis import HealthCheck, given, settings, strategies as st
    HAVE_HYP = True
except Exception:  # pragma: no cover
    HAVE_HYP = False


@pytest.mark.skipif(not HAVE_HYP, reason="hypothesis not installed")
def test_reader_matches_oracle_on_random_files(tmp_path_factory):
    d = tmp_path_factory.mktemp("hyp")

    @settings(max_examples=60, deadline=None, derandomize=True, database=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
    @given(st.integers(1, 40), st.integers(1, 300), st.integers(0, 2 ** 31 - 1), st.sampled_from([1, 2, 3, 4]), st.booleans(),
           st.sampled_from([0, 1, 2, 3]))
    def run(m, n, seed, rl, wide, nonref):
        rng = np.random.default_rng(seed)
        lim = n // opg.MAX_DIFFLIST_DIV
        g = np.zeros((m, n), dtype=np.uint8)
        vts = []
        prev = None
        for j in range(m):
            base = rng.integers(0, 4, n).astype(np.uint8) if rng.random() < 0.3 else np.full(n, rng.integers(0, 4), np.uint8)
            k = int(rng.integers(0, lim + 1)) if lim else 0
            if prev is not None and rng.random() < 0.4:
                base = prev.copy() if rng.random() < 0.5 else opg._invert(prev)
            pos = rng.choice(n, size=k, replace=False)
            base[pos] = rng.integers(0, 4, k)
            g[j] = base
            cand = [0]
            cnt = np.bincount(base, minlength=4)
            if n - np.sort(cnt)[-2:].sum() <= lim:
                cand.append(1)
            for b, t in ((0, 4), (2, 6), (3, 7)):
                if (base != b).sum() <= lim:
                    cand.append(t)
            if not base.any():
                cand.append(5)
            if prev is not None:
                if (base != prev).sum() <= lim:
                    cand.append(2)
                if (opg._invert(base) != prev).sum() <= lim:
                    cand.append(3)
            t = int(rng.choice(cand))
            vts.append(t)
            if (t & 6) != 2:
                prev = base
        path = str(d / ("h%d.pgen" % seed))
        opg.write_pgen(path, g, vts, reclen_bytes=rl, wide_vrtypes=wide, phase=wide and bool(seed & 1), nonref=nonref, seed=seed)
        o = opg.PgenOracle(path)
        with PgenFile(path) as f:
            order = rng.permutation(m)
            rows = f.read_bed_rows(order)
            for a, j in enumerate(order):
                assert (rows[a] == o.bed_row(int(j))).all(), (j, vts[j])
                assert (o.codes(int(j)) == g[j]).all()
        os.remove(path)

    run()
