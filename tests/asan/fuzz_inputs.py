"""Damaged inputs under AddressSanitizer / UBSan (test infrastructure; not part of the default suites -- minutes of g++ and thousands of runs):

  python tests/asan/fuzz_inputs.py bgen|pgen|pgenhard N [seed0]    the .bgen / .pgen readers behind include/rg_bgen.h / rg_pgen.h (csrc/bgen_api.cpp, pgen_api.cpp,
                                                                  inflate_fast.h): a small valid file with bytes flipped, runs overwritten, 32-bit fields made huge,
                                                                  the file cut short; every variant is read through every entry (read_all.cpp)
  python tests/asan/fuzz_inputs.py text N [seed0]                 the driver's text parsers (.pheno / .covar / .fam / .bim / --keep / --extract lists) through
                                                                  parse_args + read_bim_fam + read_pheno_cov of regenie_amd/host (host_prep_main.cpp)
  python tests/asan/fuzz_inputs.py loco N [seed0]                 --step 2: the prediction list and the .loco files of Step 1 (blup_read)

A run must end with exit code 0 -- the input read, or refused with an error -- within 60 s, with no sanitizer report.  Round 5: 10,000 reader and 6,100 text
mutations; three findings, all "a damaged count drives an allocation before it is checked against the file size" (.pgen variant count: minutes in
rg_pgen_open; .bgen variant count / sample-block length / inflated length: gigabytes requested), fixed in csrc/pgen_reader.h and csrc/bgen_reader.h."""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.util import synth_dosages, write_plink, write_synth_bgen, write_synth_pgen      # noqa: E402

SAN = ["-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-w"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1")
CSRC, HOST = os.path.join(ROOT, "regenie_amd", "csrc"), os.path.join(ROOT, "regenie_amd", "host")


def build(work, which):
    exe = os.path.join(work, which)
    if which == "read_all":
        cmd = ["g++"] + SAN + ["-I" + os.path.join(ROOT, "include"), os.path.join(HERE, "read_all.cpp"), os.path.join(CSRC, "bgen_api.cpp"), os.path.join(CSRC, "pgen_api.cpp")]
    else:
        cmd = ["g++"] + SAN + ["-I" + HOST, os.path.join(HERE, "host_prep_main.cpp")] + [os.path.join(HOST, f) for f in ("driver_common.cpp", "driver_inputs.cpp", "driver_models.cpp")] + \
              [os.path.join(CSRC, "bgen_api.cpp"), os.path.join(CSRC, "pgen_api.cpp")]
    r = subprocess.run(cmd + ["-o", exe, "-lz", "-lpthread"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def run(cmd):
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=60, env=ENV)
        return r.returncode, r.stderr
    except subprocess.TimeoutExpired:
        return -999, "TIMEOUT (60 s)"


TOKENS = [b"NA", b"nan", b"-", b"1e999", b"0x1p3", b"\t", b"  ", b"\r", b"", b"\n\n", b"\x00", b"1.5.2", b"+", b"e", b"FID", b"-9", b"inf", b",", b"FID_IID", b"0_0", b" 7_7 "]


def mutate_text(d, rng, head=False):
    d = bytearray(d)
    for _ in range(int(rng.integers(1, 4))):
        mode = int(rng.integers(0, 5))
        p = int(rng.integers(0, max(1, min(len(d), 400)))) if head and rng.random() < 0.5 else int(rng.integers(0, max(1, len(d))))
        if mode == 0:
            d[p:p + 1] = bytes([int(rng.integers(0, 256))])
        elif mode == 1:
            del d[p:p + int(rng.integers(1, 40))]
        elif mode == 2:
            d[p:p] = TOKENS[int(rng.integers(0, len(TOKENS)))]
        elif mode == 3:
            d = d[:p]
        else:
            q = d.find(b" ", p)
            if q >= 0:
                d[q:q + 1] = TOKENS[int(rng.integers(0, len(TOKENS)))]
    return bytes(d)


def mutate_binary(data, rng):
    d = data.copy()
    mode = int(rng.integers(0, 5))
    if mode == 0:      # a few bytes anywhere
        for _ in range(int(rng.integers(1, 6))):
            d[rng.integers(0, d.size)] = rng.integers(0, 256)
    elif mode == 1:    # the header
        for _ in range(int(rng.integers(1, 4))):
            d[rng.integers(0, min(64, d.size))] = rng.integers(0, 256)
    elif mode == 2:    # cut short
        d = d[: int(rng.integers(1, d.size))]
    elif mode == 3:    # a 32-bit field becomes huge / tiny
        p = int(rng.integers(0, d.size - 4))
        d[p:p + 4] = np.frombuffer(np.uint32(rng.choice([0, 1, 0x7fffffff, 0xffffffff, 0x10000])).tobytes(), np.uint8)
    else:              # a run of bytes
        p, L = int(rng.integers(0, d.size - 1)), int(rng.integers(1, 200))
        d[p:p + L] = rng.integers(0, 256, min(L, d.size - p))
    return d


def main():
    kind, n_iter = sys.argv[1], int(sys.argv[2])
    seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    work = tempfile.mkdtemp(prefix="rg_asan_")
    bad = 0
    if kind in ("bgen", "pgen", "pgenhard"):
        exe = build(work, "read_all")
        g = synth_dosages(40, 57, miss_rate=0.03, seed=5)
        base = os.path.join(work, "b")
        if kind == "bgen":
            write_synth_bgen(base, g, [1] * 20 + [2] * 20, seed=3)
            src, ext = base + ".bgen", ".bgen"
        else:
            write_synth_pgen(base, g, [1] * 20 + [2] * 20, seed=3, soft=0.4 if kind == "pgen" else 0.0)
            src, ext = base + ".pgen", ".pgen"
        data = np.frombuffer(open(src, "rb").read(), np.uint8)
        for it in range(n_iter):
            f = os.path.join(work, "m" + ext)
            open(f, "wb").write(mutate_binary(data, np.random.default_rng(seed0 + it)).tobytes())
            rc, err = run([exe, "bgen" if kind == "bgen" else "pgen", f])
            if rc != 0:
                bad += 1
                shutil.copy(f, os.path.join(tempfile.gettempdir(), "rg_asan_bad_%s_%d" % (kind, seed0 + it)))
                print("iteration", seed0 + it, "exit", rc, "|", "\n".join(err.strip().splitlines()[:14])[:1500], flush=True)
    else:
        exe = build(work, "host_prep")
        S = os.path.join(work, "s")
        g = synth_dosages(30, 60, miss_rate=0.02, seed=4)
        write_plink(S, g, [1] * 15 + [2] * 15, P=2, seed=4, binary=True, missing_pheno=0.05)
        lines = open(S + ".covar").read().splitlines()
        open(S + ".covar", "w").write("\n".join([lines[0] + " CAT"] + [ln + " %d" % (i % 3) for i, ln in enumerate(lines[1:])]) + "\n")
        open(S + ".keep", "w").write("".join("%d %d\n" % (i, i) for i in range(1, 50)))
        open(S + ".snps", "w").write("".join("s%d\n" % i for i in range(0, 25)))
        if kind == "loco":
            from oracle import regenie_step1 as orc
            orc.run_step1(orc.Step1Options(bed=S, pheno_file=S + ".pheno", covar_file=S + ".covar", bsize=10, bt=True, loocv=True, out=os.path.join(work, "out")), write_files=True)
            names = [os.path.join(work, f) for f in ("out_pred.list", "out_1.loco", "out_2.loco")]
        else:
            names = [S + e for e in (".pheno", ".covar", ".fam", ".bim", ".keep", ".snps")]
        orig = {f: open(f, "rb").read() for f in names}
        for it in range(n_iter):
            rng = np.random.default_rng(seed0 + it)
            for f, d in orig.items():
                open(f, "wb").write(d)
            f = names[int(rng.integers(0, len(names)))]
            open(f, "wb").write(mutate_text(orig[f], rng, head=kind == "loco"))
            if kind == "loco":
                args = ["--step", "2", "--bt", "--bed", S, "--phenoFile", S + ".pheno", "--covarFile", S + ".covar", "--bsize", "200", "--pred", names[0], "--out", os.path.join(work, "o")]
            else:
                args = ["--step", "1", "--bed", S, "--phenoFile", S + ".pheno", "--covarFile", S + ".covar", "--bsize", "10", "--out", os.path.join(work, "o"), "--catCovarList", "CAT",
                        "--keep", S + ".keep", "--extract", S + ".snps"] + (["--bt"] if it % 2 else []) + (["--apply-rint"] if it % 4 == 0 else [])
            rc, err = run([exe] + args)
            if rc != 0:
                bad += 1
                shutil.copy(f, os.path.join(tempfile.gettempdir(), "rg_asan_bad_%s_%d_%s" % (kind, seed0 + it, os.path.basename(f))))
                print("iteration", seed0 + it, "file", os.path.basename(f), "exit", rc, "|", "\n".join(err.strip().splitlines()[:16])[:1800], flush=True)
    print("%s: %d mutations, %d not clean" % (kind, n_iter, bad))
    shutil.rmtree(work, ignore_errors=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
