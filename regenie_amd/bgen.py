"""BGEN v1.2 input for the Step-1 path (`regenie --step 1 --bgen FILE`): ctypes wrapper over include/rg_bgen.h
(regenie_amd/csrc/bgen_reader.h).  Dosage rows go to Step1Engine.l0_blocks_f64_host.  No Python decode path."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .engine import RgError, load_library


class BgenFile:
    def __init__(self, path: str, threads: int = 1):
        self.lib = load_library()
        self.h = C.c_void_p()
        rc = self.lib.rg_bgen_open(C.byref(self.h), path.encode())
        if rc != 0:
            msg = self.lib.rg_bgen_last_error(self.h).decode() if self.h else "rg_bgen_open failed"
            self.close()
            raise RgError(rc, msg)
        ns, nv, cp, si = C.c_int64(), C.c_int64(), C.c_int32(), C.c_int32()
        self.lib.rg_bgen_info(self.h, C.byref(ns), C.byref(nv), C.byref(cp), C.byref(si))
        self.n_samples, self.n_variants, self.compression, self.has_sample_ids = ns.value, nv.value, cp.value, bool(si.value)
        if threads != 1:
            self._check(self.lib.rg_bgen_set_threads(self.h, int(threads)))

    def close(self) -> None:
        if getattr(self, "h", None):
            self.lib.rg_bgen_close(self.h)
        self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        self.close()

    def _check(self, rc: int) -> None:
        if rc != 0:
            raise RgError(rc, self.lib.rg_bgen_last_error(self.h).decode())

    def sample_ids(self):
        out = []
        p = C.c_char_p()
        for i in range(self.n_samples if self.has_sample_ids else 0):
            self._check(self.lib.rg_bgen_sample_id(self.h, i, C.byref(p)))
            out.append(p.value.decode())
        return out

    def variant(self, j: int):
        ch, rs, a0, a1 = C.c_char_p(), C.c_char_p(), C.c_char_p(), C.c_char_p()
        pos, off = C.c_uint32(), C.c_int64()
        self._check(self.lib.rg_bgen_variant(self.h, j, C.byref(ch), C.byref(pos), C.byref(rs), C.byref(a0), C.byref(a1), C.byref(off)))
        return dict(chrom=ch.value.decode(), pos=pos.value, rsid=rs.value.decode(), a0=a0.value.decode(), a1=a1.value.decode(), offset=off.value)

    def read_dosages(self, variant_idx, ref_first: bool = False) -> np.ndarray:
        """float64 [len(idx), n_samples]: G = prob1 + 2 prob0 (prob1 + 2 prob2 with ref_first), -3 = missing."""
        idx = np.ascontiguousarray(variant_idx, dtype=np.int64)
        rows = np.empty((idx.size, self.n_samples), dtype=np.float64)
        self._check(self.lib.rg_bgen_read_dosages(self.h, idx.size, idx.ctypes.data, 1 if ref_first else 0, rows.ctypes.data, self.n_samples))
        return rows

    def read_blocks(self, variant_idx) -> np.ndarray:
        """uint8 [len(idx), 10 + 3 n_samples]: the inflated, checked probability blocks (rg_bgen_read_blocks) -- ploidy / missingness bytes
        at [8, 8 + N), the (prob0, prob1) byte pairs at [10 + N, 10 + 3 N)."""
        idx = np.ascontiguousarray(variant_idx, dtype=np.int64)
        nb = C.c_int64()
        self._check(self.lib.rg_bgen_block_bytes(self.h, C.byref(nb)))
        blocks = np.empty((idx.size, nb.value), dtype=np.uint8)
        self._check(self.lib.rg_bgen_read_blocks(self.h, idx.size, idx.ctypes.data, blocks.ctypes.data, nb.value, 0))
        return blocks
