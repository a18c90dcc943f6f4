"""TEST INFRASTRUCTURE -- CPU restatement of the PLINK2 .pgen hardcall reader regenie uses for
`--pgen` input, plus a small .pgen WRITER used only to manufacture
synthetic fixtures that exercise every record type.

Only tests/ may import this module.  The product reader is regenie_amd/csrc/pgen_reader.h.

The algorithm lives in the reference's vendored pgenlib (external_libs/pgenlib/include/
pgenlib_read.cc); each function cites the lines it follows.  Parity is PINNED two ways:
  * against the reference's own fixture pair example.pgen / example.bed (the reference's test
    test/test_bash.sh:411-433 requires both to give byte-identical results), and
  * against the reference's reader itself, compiled from its sources by oracle/Makefile into
    oracle/_ref/libpgen_ref.so (tests/test_pgen.py: every synthetic file written here is read back
    by the real pgenlib and must give the genotypes it was written from).

What regenie asks of the file (Geno.cpp:1071-1103, :1793-1798): biallelic variants only,
ReadHardcalls(.., allele_idx=1) -> ALT-allele counts 0/1/2 and -3 for missing; files that carry a
dosage track switch regenie to Read() (dosages) -- that mode is outside the 2-bit GPU path and both
this oracle and the product reader refuse such files.
"""
from __future__ import annotations

import struct

import numpy as np

VBLOCK = 65536          # kPglVblockSize (pgenlib_misc.h:628)
DIFFLIST_GROUP = 64     # kPglDifflistGroupSize
MAX_DIFFLIST_DIV = 8    # kPglMaxDifflistLenDivisor

# pgen code (0 hom-REF, 1 het, 2 hom-ALT, 3 missing) -> PLINK1 .bed code (00 hom-A1(ALT), 01 missing,
# 10 het, 11 hom-A2(REF)); PgrPlink1ToPlink2InplaceUnsafe is the inverse map.
PGEN_TO_BED = np.array([3, 2, 0, 1], dtype=np.uint8)
HARDCALL = np.array([0.0, 1.0, 2.0, -3.0])  # kGenoRDoublePairs (pgenlibr.cpp:320)


class PgenError(ValueError):
    pass


def _vint(buf: bytes, pos: int):
    """GetVint31: LEB128, 7 bits per byte, low group first."""
    val = 0
    shift = 0
    while True:
        if pos >= len(buf):
            raise PgenError("malformed .pgen record (varint runs past the record)")
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not (b & 0x80):
            return val, pos
        shift += 7
        if shift > 28:
            raise PgenError("malformed .pgen record (varint too long)")


def _sample_id_bytes(n: int) -> int:
    """BytesToRepresentNzU32(raw_sample_ct): bsr(n)/8 + 1."""
    return (int(n).bit_length() - 1) // 8 + 1


def _unpack2(raw: bytes, n: int) -> np.ndarray:
    a = np.frombuffer(raw, dtype=np.uint8)
    out = np.empty((a.size, 4), dtype=np.uint8)
    for k in range(4):
        out[:, k] = (a >> (2 * k)) & 3
    return out.reshape(-1)[:n].copy()


class PgenOracle:
    """Header parse = PgfiInitPhase1/Phase2 (pgenlib_read.cc:684-975, :1094-1640)."""

    def __init__(self, path: str):
        self.path = path
        with open(path, "rb") as fh:
            self.data = fh.read()
        d = self.data
        if len(d) < 12 or d[0] != 0x6C or d[1] != 0x1B:
            raise PgenError("not a .pgen file (magic number mismatch)")
        mode = d[2]
        if mode == 0x01:
            raise PgenError("this is a PLINK1 .bed file; pass it with --bed")
        if mode in (0x03, 0x04):
            raise PgenError("pgen file carries dosages; only hardcall pgen files are served")
        if mode not in (0x02, 0x10, 0x11):
            raise PgenError("unsupported .pgen storage mode 0x%02x" % mode)
        self.mode = mode
        self.m, self.n = struct.unpack_from("<II", d, 3)
        ctrl = d[11]
        self.ctrl = ctrl
        n, m = self.n, self.m
        self.bpr = (n + 3) // 4
        self.max_alleles = 2
        self.dosage_present = False
        self.phase_present = False
        if mode == 0x02:  # fixed-width 2-bit records (pgenlib_read.cc:870-906)
            if ctrl & 63:
                raise PgenError("malformed .pgen header (fixed-width mode with a variable-width control byte)")
            off = 12 + ((m + 7) // 8 if (ctrl >> 6) == 3 else 0)
            if off + m * self.bpr != len(d):
                raise PgenError("unexpected .pgen file size")
            self.vrtypes = np.zeros(m + 1, dtype=np.uint8)
            self.fpos = off + np.arange(m + 1, dtype=np.int64) * self.bpr
            return
        store = ctrl & 15
        if store & 8:
            raise PgenError("unsupported .pgen header (compact single-sample vrtype modes)")
        ac_bytes = (ctrl >> 4) & 3
        if ac_bytes:  # PgenReader::Load exits on the allele-count bytes, biallelic or not (pgenlibr.cpp:65-68)
            raise PgenError("Storing of allele count information is not supported (only bi-allelic variants should be present).")
        nonref_stored = (ctrl >> 6) == 3
        nblk = (m + VBLOCK - 1) // VBLOCK
        pos = 12
        (fpos,) = struct.unpack_from("<Q", d, pos)  # only the first block offset is read (:1200-1212)
        pos += 8 * nblk
        reclen_bytes = 1 + (store & 3)
        vrtypes = np.zeros(m + 1, dtype=np.uint8)  # trailing zero: "is the next variant LD-compressed" reads one past
        fp = np.zeros(m + 1, dtype=np.int64)
        v0 = 0
        for b in range(nblk):
            cnt = min(VBLOCK, m - v0)
            if store < 4:  # 4-bit vrtypes (:1341-1350)
                nb = (cnt + 1) // 2
                a = np.frombuffer(d, dtype=np.uint8, count=nb, offset=pos)
                t = np.empty(nb * 2, dtype=np.uint8)
                t[0::2] = a & 15
                t[1::2] = a >> 4
                vrtypes[v0:v0 + cnt] = t[:cnt]
                pos += nb
            else:
                vrtypes[v0:v0 + cnt] = np.frombuffer(d, dtype=np.uint8, count=cnt, offset=pos)
                pos += cnt
            raw = np.frombuffer(d, dtype=np.uint8, count=cnt * reclen_bytes, offset=pos).reshape(cnt, reclen_bytes)
            pos += cnt * reclen_bytes
            lens = np.zeros(cnt, dtype=np.int64)
            for k in range(reclen_bytes):
                lens |= raw[:, k].astype(np.int64) << (8 * k)
            fp[v0:v0 + cnt] = fpos + np.concatenate([[0], np.cumsum(lens)[:-1]])
            fpos += int(lens.sum())
            if nonref_stored:
                pos += (cnt + 7) // 8
            v0 += cnt
        fp[m] = fpos
        if pos > fp[0] or fp[m] > len(d):
            raise PgenError("invalid .pgen header")
        self.vrtypes, self.fpos = vrtypes, fp
        vt = vrtypes[:m]
        self.dosage_present = bool((vt & 0x60).any())      # kfPgenGlobalDosagePresent (:1613-1620)
        self.phase_present = bool((vt & 0x10).any())
        if (vt & 0x08).any():
            self.max_alleles = max(self.max_alleles, 3)
        self._ld_vidx = -1
        self._ld_geno = None

    # ---- record decoding ---------------------------------------------------------------------
    def _difflist(self, rec: bytes, pos: int):
        """ParseAndSaveDifflist (pgenlib_read.cc:2177-2267): returns (sample ids, 2-bit codes, pos)."""
        n = self.n
        ln, pos = _vint(rec, pos)
        if ln == 0:
            return np.zeros(0, np.int64), np.zeros(0, np.uint8), pos
        if ln > n // MAX_DIFFLIST_DIV:
            raise PgenError("malformed .pgen record (difflist too long)")
        groups = (ln + DIFFLIST_GROUP - 1) // DIFFLIST_GROUP
        sib = _sample_id_bytes(n)
        first = [int.from_bytes(rec[pos + g * sib: pos + (g + 1) * sib], "little") for g in range(groups)]
        pos += groups * (sib + 1) - 1
        rg_bytes = (ln + 3) // 4
        if pos + rg_bytes > len(rec):
            raise PgenError("malformed .pgen record (difflist runs past the record)")
        codes = _unpack2(rec[pos:pos + rg_bytes], ln)
        pos += rg_bytes
        ids = np.empty(ln, dtype=np.int64)
        k = 0
        for g in range(groups):
            cur = first[g]
            ids[k] = cur
            k += 1
            for _ in range(min(DIFFLIST_GROUP, ln - g * DIFFLIST_GROUP) - 1):
                dlt, pos = _vint(rec, pos)
                cur += dlt
                ids[k] = cur
                k += 1
            if cur >= n:
                raise PgenError("malformed .pgen record (difflist sample index out of range)")
        return ids, codes, pos

    def _ldbase_vidx(self, vidx: int) -> int:
        """GetLdbaseVidx (:1840-1860): the last earlier variant that is not LD-compressed."""
        v = vidx - 1
        while v >= 0 and (self.vrtypes[v] & 6) == 2:
            v -= 1
        if v < 0:
            raise PgenError("malformed .pgen file (LD-compressed variant without a base)")
        return v

    def codes(self, vidx: int) -> np.ndarray:
        """ReadGenovecSubsetUnsafe (:2837-2900) without subsetting: pgen 2-bit codes of one variant."""
        return self._codes_pos(vidx)[0]

    def _codes_pos(self, vidx: int):
        """(codes, offset of the first byte after the main genotype track inside the record)."""
        if not (0 <= vidx < self.m):
            raise IndexError("variant index out of range")
        vt = int(self.vrtypes[vidx]) & 7
        rec = self.data[self.fpos[vidx]:self.fpos[vidx + 1]]
        n = self.n
        if (vt & 6) == 2:  # LD-compressed: ldbase, patched by a difflist, inverted for type 3
            base = self._ldbase_vidx(vidx)
            if self._ld_vidx != base:
                self._ld_geno = self.codes(base)
                self._ld_vidx = base
            g = self._ld_geno.copy()
            ids, cd, pos = self._difflist(rec, 0)
            g[ids] = cd
            if vt == 3:  # GenovecInvertUnsafe: 0<->2
                g = np.where(g == 0, 2, np.where(g == 2, 0, g)).astype(np.uint8)
            return g, pos
        pos = 0
        if not (vt & 4):
            if vt & 3:  # ParseOnebitUnsafe (:2597-2680)
                nb = (n + 7) // 8
                if len(rec) < 1 + nb:
                    raise PgenError("malformed .pgen record (onebit track runs past the record)")
                c2 = rec[0]
                lo, dlt = c2 >> 2, c2 & 3
                bits = np.unpackbits(np.frombuffer(rec, dtype=np.uint8, count=nb, offset=1), bitorder="little")[:n]
                g = (lo + dlt * bits).astype(np.uint8)
                ids, cd, pos = self._difflist(rec, 1 + nb)
                g[ids] = cd
            else:
                if len(rec) < self.bpr:
                    raise PgenError("malformed .pgen record (2-bit track runs past the record)")
                g = _unpack2(rec[:self.bpr], n)
                pos = self.bpr
        elif (vt & 3) == 1:  # all hom-REF, empty record (:2727-2729)
            g = np.zeros(n, dtype=np.uint8)
        else:
            g = np.full(n, vt & 3, dtype=np.uint8)
            ids, cd, pos = self._difflist(rec, 0)
            g[ids] = cd
        return g, pos

    def dosages(self, vidx: int) -> np.ndarray:
        """What PgenReader::Read(.., allele_idx=1) returns (pgenlibr.cpp:323-349 -> PgrGet1D, pgenlib_read.cc:7459-7497,
        ParseDosage16 :7185-7330, Dosage16ToDoubles): ALT dosage = value / 16384 where the variant stores one for the
        sample, the hardcall (0/1/2, -3 missing) elsewhere.  Biallelic files only."""
        g, pos = self._codes_pos(vidx)
        out = HARDCALL[g].copy()
        vt = int(self.vrtypes[vidx])
        if not (vt & 0x60):
            return out
        if vt & 0x08:
            raise PgenError("multiallelic variant")
        rec = self.data[self.fpos[vidx]:self.fpos[vidx + 1]]
        n = self.n
        if vt & 0x10:  # SkipAux2 (:6819-6840): hardcall-phase track, sized by the number of heterozygous calls
            het = int((g == 1).sum())
            first = 1 + het // 8
            if het == 0 or pos + first > len(rec):       # ParseAux2Subset (:6743-6760): no het, no phase track
                raise PgenError("malformed .pgen record (phase track)")
            if rec[pos] & 1:
                present = int(np.unpackbits(np.frombuffer(rec, np.uint8, first, pos)).sum()) - 1
                if present == 0:
                    raise PgenError("malformed .pgen record (phase track without a phased call)")
                first += (present + 7) // 8
            pos += first
        kind = vt & 0x60
        if kind == 0x20:      # list of sample ids (a difflist without replacement codes), then one value per entry
            ids, pos = self._deltalist(rec, pos)
        elif kind == 0x40:    # one value per sample; 65535 = none
            ids = np.arange(n)
        else:                 # one bit per sample, then one value per set bit
            nb = (n + 7) // 8
            if pos + nb > len(rec):
                raise PgenError("malformed .pgen record (dosage bit array runs past the record)")
            ids = np.flatnonzero(np.unpackbits(np.frombuffer(rec, np.uint8, nb, pos), bitorder="little")[:n])
            pos += nb
        if pos + 2 * ids.size > len(rec):
            raise PgenError("malformed .pgen record (dosage values run past the record)")
        vals = np.frombuffer(rec, dtype="<u2", count=ids.size, offset=pos)
        keep = vals != 65535 if kind == 0x40 else np.ones(ids.size, bool)
        out[ids[keep]] = vals[keep].astype(np.float64) / 16384.0
        return out

    def _deltalist(self, rec: bytes, pos: int):
        """ParseAndSaveDeltalistAsBitarr: the difflist header and id stream without the 2-bit codes."""
        n = self.n
        ln, pos = _vint(rec, pos)
        if ln == 0:
            return np.zeros(0, np.int64), pos
        if ln > n // MAX_DIFFLIST_DIV:
            raise PgenError("malformed .pgen record (dosage list too long)")
        groups = (ln + DIFFLIST_GROUP - 1) // DIFFLIST_GROUP
        sib = _sample_id_bytes(n)
        first = [int.from_bytes(rec[pos + g * sib: pos + (g + 1) * sib], "little") for g in range(groups)]
        pos += groups * (sib + 1) - 1
        ids = np.empty(ln, dtype=np.int64)
        k = 0
        for g in range(groups):
            cur = first[g]
            ids[k] = cur
            k += 1
            for _ in range(min(DIFFLIST_GROUP, ln - g * DIFFLIST_GROUP) - 1):
                dlt, pos = _vint(rec, pos)
                cur += dlt
                ids[k] = cur
                k += 1
            if cur >= n:
                raise PgenError("malformed .pgen record (dosage list sample index out of range)")
        return ids, pos

    def hardcalls(self, vidx: int) -> np.ndarray:
        """What PgenReader::ReadHardcalls(.., allele_idx=1) returns (pgenlibr.cpp:296-321)."""
        return HARDCALL[self.codes(vidx)]

    def bed_row(self, vidx: int) -> np.ndarray:
        """The same genotypes in PLINK1 .bed coding, ceil(N/4) bytes, padding bits zero."""
        b = PGEN_TO_BED[self.codes(vidx)]
        pad = np.zeros(self.bpr * 4, dtype=np.uint8)
        pad[:self.n] = b
        q = pad.reshape(-1, 4)
        return (q[:, 0] | (q[:, 1] << 2) | (q[:, 2] << 4) | (q[:, 3] << 6)).astype(np.uint8)


# ---- writer (fixtures only) --------------------------------------------------------------------
def _enc_vint(v: int) -> bytes:
    out = bytearray()
    while v >= 128:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _pack2(codes: np.ndarray) -> bytes:
    n = codes.size
    pad = np.zeros((n + 3) // 4 * 4, dtype=np.uint8)
    pad[:n] = codes
    q = pad.reshape(-1, 4)
    return (q[:, 0] | (q[:, 1] << 2) | (q[:, 2] << 4) | (q[:, 3] << 6)).astype(np.uint8).tobytes()


def _enc_difflist(ids: np.ndarray, codes: np.ndarray, n: int) -> bytes:
    ln = int(ids.size)
    out = bytearray(_enc_vint(ln))
    if ln == 0:
        return bytes(out)
    if ln > n // MAX_DIFFLIST_DIV:
        raise ValueError("difflist too long for this record type (%d > %d)" % (ln, n // MAX_DIFFLIST_DIV))
    sib = _sample_id_bytes(n)
    groups = (ln + DIFFLIST_GROUP - 1) // DIFFLIST_GROUP
    deltas = []
    for g in range(groups):
        lo, hi = g * DIFFLIST_GROUP, min(ln, (g + 1) * DIFFLIST_GROUP)
        out += int(ids[lo]).to_bytes(sib, "little")
        deltas.append(b"".join(_enc_vint(int(ids[k] - ids[k - 1])) for k in range(lo + 1, hi)))
    for g in range(groups - 1):
        out.append(len(deltas[g]) - 63)
    out += _pack2(codes)
    for dl in deltas:
        out += dl
    return bytes(out)


def _invert(g: np.ndarray) -> np.ndarray:
    return np.where(g == 0, 2, np.where(g == 2, 0, g)).astype(np.uint8)


def encode_record(g: np.ndarray, vt: int, ldbase: np.ndarray | None) -> bytes:
    """Main genotype track of one variant in record type vt (0..7); g holds pgen codes."""
    n = g.size
    if vt == 0:
        return _pack2(g)
    if vt == 1:
        cnt = np.bincount(g, minlength=4)
        a, b = sorted(np.argsort(-cnt, kind="stable")[:2].tolist())
        bits = np.packbits((g == b).astype(np.uint8), bitorder="little").tobytes()
        rare = np.flatnonzero((g != a) & (g != b))
        return bytes([a * 4 + (b - a)]) + bits + _enc_difflist(rare, g[rare], n)
    if vt in (2, 3):
        if ldbase is None:
            raise ValueError("LD-compressed record without a base variant")
        tgt = _invert(g) if vt == 3 else g
        diff = np.flatnonzero(tgt != ldbase)
        return _enc_difflist(diff, tgt[diff], n)
    if vt == 5:
        if g.any():
            raise ValueError("record type 5 needs an all hom-REF variant")
        return b""
    base = vt & 3
    diff = np.flatnonzero(g != base)
    return _enc_difflist(diff, g[diff], n)


def write_pgen(path: str, geno: np.ndarray, vrtypes, *, reclen_bytes: int = 2, wide_vrtypes: bool = False,
               phase: bool = False, nonref: int = 0, dosage_variant: int | None = None, allele_counts=None,
               mode: int = 0x10, seed: int = 0, dosage: dict | None = None) -> None:
    """Writes a mode-0x10 .pgen.  geno: M x N pgen codes; vrtypes: M record types (0..7).
    phase=True / "explicit" (needs wide_vrtypes) appends a hardcall-phase track to every variant that has hets, so a
    hardcall reader must step over it; dosage_variant marks one variant as carrying a dosage track;
    allele_counts (M values) adds the per-variant allele-count bytes of a file that may be multiallelic."""
    rng = np.random.default_rng(seed)
    m, n = geno.shape
    if (phase or dosage_variant is not None or dosage) and not wide_vrtypes:
        raise ValueError("phase/dosage tracks need 8-bit vrtypes")
    recs, vts = [], []
    ldbase = None
    for j in range(m):
        g = geno[j].astype(np.uint8)
        vt = int(vrtypes[j])
        rec = encode_record(g, vt, ldbase)
        if (vt & 6) != 2:
            ldbase = g.copy()
        full = vt
        if phase:
            het = int((g == 1).sum())
            if het:
                full |= 0x10
                if phase == "explicit" and j % 2:
                    # first bit 1, then one "is this het phased" bit per het; the phase bits of the phased ones follow
                    pp = rng.integers(0, 2, het).astype(np.uint8)
                    pp[int(rng.integers(0, het))] = 1           # a track without any phased het is malformed (ParseAux2Subset)
                    first = np.zeros((1 + het // 8) * 8, dtype=np.uint8)
                    first[0] = 1
                    first[1:1 + het] = pp
                    rec += np.packbits(first, bitorder="little").tobytes()
                    rec += np.packbits(rng.integers(0, 2, int(pp.sum())).astype(np.uint8), bitorder="little").tobytes()
                else:  # first bit 0 = every het phased, then one phase bit per het
                    bits = np.concatenate([[0], rng.integers(0, 2, het)]).astype(np.uint8)
                    rec += np.packbits(bits, bitorder="little").tobytes()
        if dosage_variant == j:
            full |= 0x40  # unconditional dosage: a 16-bit value per sample
            rec += rng.integers(0, 32768, n).astype("<u2").tobytes()
        if dosage is not None and j in dosage:
            kind, ids, vals = dosage[j]          # kind 0x20 list / 0x40 all samples / 0x60 bit array; vals: uint16 per id
            ids = np.asarray(ids, dtype=np.int64)
            vals = np.asarray(vals, dtype=np.uint16)
            full |= kind
            if kind == 0x20:
                dl = _enc_difflist(ids, np.zeros(ids.size, np.uint8), n)
                # a deltalist is a difflist without the 2-bit codes: cut them out again
                hdr_len = len(_enc_vint(ids.size))
                if ids.size:
                    groups = (ids.size + DIFFLIST_GROUP - 1) // DIFFLIST_GROUP
                    cut = hdr_len + groups * (_sample_id_bytes(n) + 1) - 1
                    dl = dl[:cut] + dl[cut + (ids.size + 3) // 4:]
                rec += dl + vals.astype("<u2").tobytes()
            elif kind == 0x40:
                full_vals = np.full(n, 65535, dtype=np.uint16)
                full_vals[ids] = vals
                rec += full_vals.astype("<u2").tobytes()
            else:
                bits = np.zeros(n, dtype=np.uint8)
                bits[ids] = 1
                rec += np.packbits(bits, bitorder="little").tobytes() + vals.astype("<u2").tobytes()
        recs.append(rec)
        vts.append(full)
    nblk = (m + VBLOCK - 1) // VBLOCK
    store = (4 if wide_vrtypes else 0) | (reclen_bytes - 1)
    ctrl = store | (nonref << 6) | ((1 << 4) if allele_counts is not None else 0)
    hdr = bytearray(b"\x6c\x1b" + bytes([mode]) + struct.pack("<II", m, n) + bytes([ctrl]))
    blocks = bytearray()
    for b in range(nblk):
        lo, hi = b * VBLOCK, min(m, (b + 1) * VBLOCK)
        vt = np.array(vts[lo:hi], dtype=np.uint8)
        if wide_vrtypes:
            blocks += vt.tobytes()
        else:
            pad = np.zeros((vt.size + 1) // 2 * 2, dtype=np.uint8)
            pad[:vt.size] = vt
            blocks += (pad[0::2] | (pad[1::2] << 4)).astype(np.uint8).tobytes()
        for j in range(lo, hi):
            if len(recs[j]) >= 1 << (8 * reclen_bytes):
                raise ValueError("record does not fit the record-length width")
            blocks += len(recs[j]).to_bytes(reclen_bytes, "little")
        if allele_counts is not None:
            blocks += np.asarray(allele_counts[lo:hi], dtype=np.uint8).tobytes()
        if nonref == 3:
            blocks += rng.integers(0, 256, (hi - lo + 7) // 8).astype(np.uint8).tobytes()
    if mode == 0x11:  # "extensions present": two empty extension-type sets sit between the header and the records
        blocks += b"\x00\x00"
    first = 12 + 8 * nblk + len(blocks)
    offs = []
    pos = first
    for b in range(nblk):
        offs.append(pos)
        pos += sum(len(r) for r in recs[b * VBLOCK:(b + 1) * VBLOCK])
    with open(path, "wb") as fh:
        fh.write(bytes(hdr))
        fh.write(struct.pack("<%dQ" % nblk, *offs))
        fh.write(bytes(blocks))
        for r in recs:
            fh.write(r)


def write_pgen_fixed(path: str, geno: np.ndarray) -> None:
    """Mode 0x02: fixed-width 2-bit records."""
    m, n = geno.shape
    with open(path, "wb") as fh:
        fh.write(b"\x6c\x1b\x02" + struct.pack("<II", m, n) + b"\x40")
        for j in range(m):
            fh.write(_pack2(geno[j].astype(np.uint8)))


def write_pvar_psam(prefix: str, chroms, n: int) -> None:
    with open(prefix + ".pvar", "w") as fh:
        fh.write("#CHROM\tPOS\tID\tREF\tALT\n")
        for j, c in enumerate(chroms):
            fh.write("%d\t%d\tv%d\tA\tC\n" % (c, j + 1, j + 1))
    with open(prefix + ".psam", "w") as fh:
        fh.write("#FID\tIID\tSEX\n")
        for i in range(n):
            fh.write("%d\t%d\tNA\n" % (i + 1, i + 1))
