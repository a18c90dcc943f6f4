"""Builds librg_step1_hip.so (gfx950) in-tree with hipcc.  No JIT cache: the .so travels with the repo
snapshot to the GPU box (it is git-ignored, not gpurun-ignored)."""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "librg_step1_hip.so")
SOURCES = ["rg_api.hip", "bed_prep.hip", "gram_i8.hip", "gram_fp4.hip", "assemble.hip", "chol.hip", "pred.hip", "l1.hip", "ubench.hip", "loocv.hip", "loocv_tri.hip", "l1x.hip", "l0_f64.hip", "step2_qt.hip", "step2_bt.hip", "rg_group.hip", "pred_i8.hip", "xy_i8.hip", "wgram_bf16.hip", "bgen_inflate.hip",
           "pgen_api.cpp", "bgen_api.cpp"]  # host-only: .pgen input (include/rg_pgen.h), BGEN v1.2 input (include/rg_bgen.h)
HOST_SOURCES = ["driver_common.cpp", "driver_models.cpp", "driver_inputs.cpp", "driver_step2.cpp", "driver_step1.cpp", "driver_main.cpp"]  # regenie-amd (host/driver.h)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-unused-value", "-Wno-unused-result", "-Wno-inline-asm"]


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _digest(paths) -> str:
    """Digest of the sources by content and by their path RELATIVE to the package: the same tree gives the same stamp wherever it is checked
    out (the GPU box runs a copy under another root; profiles/*traffic*.json are keyed on this stamp)."""
    h = hashlib.sha256()
    for p in sorted(os.path.relpath(os.path.realpath(q), os.path.realpath(HERE)) for q in paths):
        with open(os.path.join(HERE, p), "rb") as fh:
            h.update(p.encode())
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def source_digest() -> str:
    """The stamp build() writes to lib/build.stamp for the current sources."""
    return _digest(_deps())


def library_digest() -> str:
    """The digest of what librg_step1_hip.so is built from (the kernels and the C ABI; NOT the host driver above it): build() writes it to
    lib/library.stamp, and the PMC traffic files under profiles/ are keyed on it -- a change to host/*.cpp does not change what a kernel moves."""
    return _digest(_lib_deps())


def kernel_file_digests() -> dict:
    """sha256 per source file of the kernel library (path relative to the package) + of the compile flags: the PMC traffic files record
    them, and bench.py accepts a file for a kernel group as long as the files that define and launch THAT group are unchanged (a change to
    bgen_inflate.hip does not move what the Cholesky kernels read)."""
    out = {}
    for q in _lib_deps():
        rel = os.path.relpath(os.path.realpath(q), os.path.realpath(HERE))
        with open(os.path.join(HERE, rel), "rb") as fh:
            out[rel] = hashlib.sha256(fh.read()).hexdigest()
    out["flags"] = hashlib.sha256(" ".join(FLAGS).encode()).hexdigest()
    return out


def _deps():
    return _lib_deps() + [os.path.join(HERE, "host", f) for f in HOST_SOURCES + ["driver.h", "fmt_g6.h"]]


def _lib_deps():
    return [os.path.join(CSRC, s) for s in SOURCES] + [
                                                      os.path.join(CSRC, "rg_internal.h"),
                                                      os.path.join(CSRC, "chol_common.h"),
                                                      os.path.join(CSRC, "chol_p128.h"),
                                                      os.path.join(CSRC, "step2_internal.h"),
                                                      os.path.join(CSRC, "pgen_reader.h"),
                                                      os.path.join(CSRC, "bgen_reader.h"),
                                                      os.path.join(CSRC, "inflate_fast.h"),
                                                      os.path.join(HERE, "..", "include", "rg_bgen.h"),
                                                      os.path.join(HERE, "..", "include", "rg_pgen.h"),
                                                      os.path.join(HERE, "..", "include", "rg_step1.h"),
                                                      os.path.join(HERE, "..", "include", "rg_step2.h")]


def _write_file_digests():
    import json
    with open(os.path.join(LIBDIR, "kernel_files.json"), "w") as fh:
        json.dump(kernel_file_digests(), fh, indent=0, sort_keys=True)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "build.stamp")
    dig = source_digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        with open(os.path.join(LIBDIR, "library.stamp"), "w") as fh:      # (a tree built before this file existed)
            fh.write(library_digest())
        _write_file_digests()
        return LIB
    hipcc = _hipcc()
    objs = []

    def compile_one(src):
        obj = os.path.join(LIBDIR, os.path.splitext(src)[0] + ".o")
        flags = FLAGS if src.endswith(".hip") else [f for f in FLAGS if not f.startswith("--offload-arch")]
        cmd = [hipcc] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return src, obj, r

    with cf.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        for src, obj, r in ex.map(compile_one, SOURCES):
            if r.returncode != 0:
                raise RuntimeError("hipcc failed on %s:\n%s\n%s" % (src, r.stdout, r.stderr))
            if verbose and r.stderr.strip():
                sys.stderr.write(r.stderr)
            objs.append(obj)
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lz", "-ldl"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    # C++ host driver (the reference's host language): `regenie-amd --step 1 ...`
    bindir = os.path.join(HERE, "bin")
    os.makedirs(bindir, exist_ok=True)
    def compile_host(src):
        obj = os.path.join(LIBDIR, os.path.splitext(src)[0] + ".o")
        return src, obj, subprocess.run([hipcc, "-x", "c++", "-O2", "-std=c++17", "-c", os.path.join(HERE, "host", src), "-o", obj],
                                        capture_output=True, text=True)

    hobjs = []
    with cf.ThreadPoolExecutor(max_workers=len(HOST_SOURCES)) as ex:
        for src, obj, r in ex.map(compile_host, HOST_SOURCES):
            if r.returncode != 0:
                raise RuntimeError("host driver build failed on %s:\n%s\n%s" % (src, r.stdout, r.stderr))
            hobjs.append(obj)
    r = subprocess.run([hipcc] + hobjs + ["-o", os.path.join(bindir, "regenie-amd"), "-L" + LIBDIR, "-lrg_step1_hip",
                        "-Wl,-rpath,$ORIGIN/../lib", "-lz", "-lpthread"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("host driver link failed:\n%s\n%s" % (r.stdout, r.stderr))
    with open(os.path.join(LIBDIR, "library.stamp"), "w") as fh:
        fh.write(library_digest())
    _write_file_digests()
    with open(stamp, "w") as fh:
        fh.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
