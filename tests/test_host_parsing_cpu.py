"""The text-input helpers of the C++ driver (regenie_amd/host/driver_common.cpp), compiled with g++ into a small harness (no GPU involved):
  convert_double_tok   must give, bit for bit, what convert_double (= the reference's strtod-based conversion, Regenie.cpp:1663-1675) gives
                       for the same characters -- its fast path (at most 15 significant digits, decimal exponent within +-22: one exactly
                       rounded multiplication or division) and everything it hands back to strtod;
  tokenize             the whitespace-separated tokens `is >> t` would give;
  IdIndex              FID_IID -> sample index without building the key;
  usable_cpus          hardware threads worth using (affinity mask, cgroup CPU quota).
These parse every phenotype / covariate value and every LOCO prediction of a run."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = r'''
#include "driver.h"
#include <cstring>
using namespace rgdrv;
// tokens are NUL-separated in buf; out[i] / ref[i] = the two conversions (a thrown conversion: flag 1 / 2 in threw[i])
extern "C" int conv_both(const char* buf, int64_t ntok, double* out, double* ref, int* threw) {
  const char* p = buf;
  for (int64_t i = 0; i < ntok; ++i) {
    const size_t len = strlen(p);
    threw[i] = 0;
    try { out[i] = convert_double_tok(p, p + len); } catch (...) { threw[i] |= 1; out[i] = 0; }
    try { ref[i] = convert_double(std::string(p, len)); } catch (...) { threw[i] |= 2; ref[i] = 0; }
    p += len + 1;
  }
  return 0;
}
extern "C" int tok_offsets(const char* line, int64_t len, int maxtok, int32_t* beg, int32_t* end) {
  std::vector<Tok> t((size_t)maxtok);
  const int n = tokenize(line, line + len, t.data(), maxtok);
  for (int i = 0; i < n && i < maxtok; ++i) { beg[i] = (int32_t)(t[i].b - line); end[i] = (int32_t)(t[i].e - line); }
  return n;
}
extern "C" int64_t id_lookup(const char* ids_buf, int64_t nids, const char* fid, const char* iid) {
  static std::vector<std::string> ids;
  static IdIndex* ix = nullptr;
  if (ids_buf) {
    ids.clear();
    const char* p = ids_buf;
    for (int64_t i = 0; i < nids; ++i) { ids.emplace_back(p); p += ids.back().size() + 1; }
    delete ix;
    ix = new IdIndex(ids);
    return (int64_t)ids.size();
  }
  return ix->find(fid, fid + strlen(fid), iid, iid + strlen(iid));
}
extern "C" int cpus() { return usable_cpus(); }
'''


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("hostparse")
    src = d / "h.cpp"
    src.write_text(HARNESS)
    so = d / "libhostparse.so"
    host = os.path.join(ROOT, "regenie_amd", "host")
    libdir = os.path.join(ROOT, "regenie_amd", "lib")
    if not os.path.exists(os.path.join(libdir, "librg_step1_hip.so")):
        import __graft_entry__ as g
        g.build()
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I" + host, os.path.join(host, "driver_common.cpp"), str(src), "-o", str(so),
                        "-L" + libdir, "-lrg_step1_hip", "-Wl,-rpath," + libdir, "-lz", "-lpthread"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return C.CDLL(str(so))


def _conv(lib, toks):
    buf = b"\0".join(t.encode() for t in toks) + b"\0"
    n = len(toks)
    out, ref = np.zeros(n), np.zeros(n)
    threw = np.zeros(n, dtype=np.int32)
    lib.conv_both(buf, C.c_int64(n), out.ctypes.data_as(C.c_void_p), ref.ctypes.data_as(C.c_void_p), threw.ctypes.data_as(C.c_void_p))
    return out, ref, threw


def test_convert_double_tok_is_strtod_bit_for_bit(lib):
    rng = np.random.default_rng(3)
    toks = ["0", "-0", "0.0", "-0.000", "1", "-1", "5.", ".5", "-.5", "1e5", "1E-3", "1e+22", "1e22", "1e23", "1e-22", "1e-23", "123456789012345", "1234567890123456",
            "0.000000000000001", "9007199254740993", "1.7976931348623157e308", "4.9e-324", "1e-400", "1e400", "NA", "nan", "inf", "-inf", "NaN", "Inf", "infinity",
            "+1.5", "0x10", "1.5abc", "1e", "1e+", "--1", "1..2", "1.2.3", "-", ".", "e5", "abc", "", "00012.50", "-000.125e1", "3.14159265358979", "2.718281828459045",
            "0.1", "0.2", "0.3", "123.456e-7", "1e0000005", "1e1234567", "9" * 15, "9" * 16, "0." + "0" * 25 + "1", "1" + "0" * 22, "1" + "0" * 23]
    # what the files of a run hold: %.6f-style values, integers, scientific notation with 6 - 17 digits, small and large exponents
    for _ in range(20000):
        kind = rng.integers(0, 6)
        x = rng.standard_normal() * 10.0 ** rng.integers(-8, 9)
        if kind == 0:
            toks.append("%.6f" % x)
        elif kind == 1:
            toks.append("%d" % int(x * 1000))
        elif kind == 2:
            toks.append("%.*e" % (int(rng.integers(0, 17)), x))
        elif kind == 3:
            toks.append("%.*g" % (int(rng.integers(1, 18)), x))
        elif kind == 4:
            toks.append(repr(float(x)))
        else:
            m = int(rng.integers(0, 10 ** 15))
            toks.append("%s%d.%0*de%d" % ("-" if rng.random() < 0.5 else "", m // 1000, 3, m % 1000, int(rng.integers(-25, 26))))
    out, ref, threw = _conv(lib, toks)
    assert ((threw == 0) | (threw == 3)).all(), [toks[i] for i in np.nonzero((threw != 0) & (threw != 3))[0][:5]]      # both convert or both refuse
    ok = threw == 0
    a = out[ok].view(np.uint64)
    b = ref[ok].view(np.uint64)
    both_nan = np.isnan(out[ok]) & np.isnan(ref[ok])
    bad = np.nonzero((a != b) & ~both_nan)[0]
    assert bad.size == 0, [(np.array(toks)[ok][i], out[ok][i], ref[ok][i]) for i in bad[:5]]
    # the fast path gives what Python's correctly rounded float() gives
    for t in ("0.1", "123.456e-7", "2.718281828459045", "-000.125e1", "1e22"):
        i = toks.index(t)
        assert struct.pack("<d", out[i]) == struct.pack("<d", float(t))
    assert out[toks.index("NA")] == out[toks.index("nan")] == out[toks.index("inf")] == -999       # the reference's missing-value code
    assert threw[toks.index("abc")] == 3 and threw[toks.index("")] == 3


def test_tokenize_is_stream_extraction(lib):
    rng = np.random.default_rng(4)
    lines = ["", " ", "a", " a ", "a b", "a\tb  c\r", "1 2\t3\t\t4   ", "FID IID Y1 Y2", "\t\tx"]
    for _ in range(200):
        n = int(rng.integers(0, 40))
        parts = []
        for _ in range(n):
            parts.append("".join(rng.choice(list("abc123.-e_"), size=int(rng.integers(1, 9)))))
            parts.append("".join(rng.choice(list(" \t"), size=int(rng.integers(1, 4)))))
        lines.append(("".join(rng.choice(list(" \t"), size=int(rng.integers(0, 3)))) + "".join(parts)))
    for ln in lines:
        raw = ln.encode()
        beg = np.zeros(64, dtype=np.int32)
        end = np.zeros(64, dtype=np.int32)
        n = lib.tok_offsets(raw, C.c_int64(len(raw)), 64, beg.ctypes.data_as(C.c_void_p), end.ctypes.data_as(C.c_void_p))
        want = ln.split()
        assert n == len(want), ln
        assert [raw[beg[i]:end[i]].decode() for i in range(min(n, 64))] == want[:64]
    # more tokens than the caller's array: the count is still the line's
    raw = b"1 2 3 4 5 6 7 8 9 10"
    assert lib.tok_offsets(raw, C.c_int64(len(raw)), 3, beg.ctypes.data_as(C.c_void_p), end.ctypes.data_as(C.c_void_p)) == 10


def test_id_index_finds_what_the_map_finds(lib):
    lib.id_lookup.restype = C.c_int64
    rng = np.random.default_rng(5)
    ids = ["%d_%d" % (i + 1, i + 1) for i in range(5000)] + ["fam_a_b_%d" % i for i in range(300)] + ["F%d_I_%d" % (i, i) for i in range(300)] + ["a__b", "a_b"]
    buf = b"\0".join(s.encode() for s in ids) + b"\0"
    assert lib.id_lookup(buf, C.c_int64(len(ids)), None, None) == len(ids)
    where = {s: i for i, s in enumerate(ids)}
    for s in list(rng.choice(ids, size=400)) + ["a__b", "a_b"]:
        for cut in [k for k, ch in enumerate(s) if ch == "_" and 0 < k < len(s) - 1]:        # every split of the key at an underscore names the same sample (tokens are never empty)
            assert lib.id_lookup(None, 0, s[:cut].encode(), s[cut + 1:].encode()) == where[s], (s, cut)
    for fid, iid in (("0", "0"), ("5001", "5001"), ("1", "2"), ("fam_a", "b"), ("a", "_b"), ("fam_a_b", "300")):
        assert lib.id_lookup(None, 0, fid.encode(), iid.encode()) == where.get(fid + "_" + iid, -1)


def test_usable_cpus(lib):
    n = lib.cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    try:
        assert n <= len(os.sched_getaffinity(0))
    except AttributeError:
        pass
