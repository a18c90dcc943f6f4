"""PLINK2 .pgen hardcall input for the Step-1 path (`regenie --step 1 --pgen PFX`).

A thin ctypes wrapper over include/rg_pgen.h (regenie_amd/csrc/pgen_reader.h): variant records are
decoded on the host into the 2-bit PLINK1 .bed rows that `Step1Engine.l0_blocks` takes, so a pgen run
hands the GPU the same bytes as the equivalent bed run.  Mirrors the reference's PgenReader as regenie
uses it (Geno.cpp:1071-1103 prep_pgen, :1781-1798 ReadHardcalls per variant).  No Python decode path:
if the library is missing this raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .engine import RgError, load_library


class PgenFile:
    def __init__(self, path: str, threads: int = 1):
        self.lib = load_library()
        self.h = C.c_void_p()
        rc = self.lib.rg_pgen_open(C.byref(self.h), path.encode())
        if rc != 0:
            msg = self.lib.rg_pgen_last_error(self.h).decode() if self.h else "rg_pgen_open failed"
            self.close()
            raise RgError(rc, msg)
        ns, nv, ac, ph, ds = C.c_int64(), C.c_int64(), C.c_int32(), C.c_int32(), C.c_int32()
        self.lib.rg_pgen_info(self.h, C.byref(ns), C.byref(nv), C.byref(ac), C.byref(ph), C.byref(ds))
        self.n_samples, self.n_variants = ns.value, nv.value
        self.max_alleles, self.phase_present, self.dosage_present = ac.value, bool(ph.value), bool(ds.value)
        self.bytes_per_row = (self.n_samples + 3) // 4
        if threads != 1:
            self.set_threads(threads)

    def set_threads(self, n: int) -> None:
        """Worker threads for read_bed_rows (the reference decodes a block's variants under OpenMP)."""
        self._check(self.lib.rg_pgen_set_threads(self.h, int(n)))

    def close(self) -> None:
        if getattr(self, "h", None):
            self.lib.rg_pgen_close(self.h)
        self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        self.close()

    def _check(self, rc: int) -> None:
        if rc != 0:
            raise RgError(rc, self.lib.rg_pgen_last_error(self.h).decode())

    def read_bed_rows(self, variant_idx) -> np.ndarray:
        """Rows of the given variants (0-based file indices) in .bed coding: uint8 [len(idx), ceil(N/4)]."""
        if not self.h:
            raise RgError(-1, "pgen file is closed")
        idx = np.ascontiguousarray(variant_idx, dtype=np.int64)
        rows = np.empty((idx.size, self.bytes_per_row), dtype=np.uint8)
        self._check(self.lib.rg_pgen_read_bed_rows(self.h, idx.size, idx.ctypes.data, rows.ctypes.data, self.bytes_per_row))
        return rows

    def read_dosages(self, variant_idx: int) -> np.ndarray:
        """ALT dosages where the file stores them, hardcalls elsewhere, -3 for missing: PgenReader::Read(.., allele_idx=1)."""
        if not self.h:
            raise RgError(-1, "pgen file is closed")
        out = np.empty(self.n_samples, dtype=np.float64)
        self._check(self.lib.rg_pgen_read_dosages(self.h, int(variant_idx), out.ctypes.data))
        return out

    def read_dosage_rows(self, variant_idx) -> np.ndarray:
        """float64 [len(idx), n_samples]: read_dosages for a block of variants, spread over the worker threads."""
        if not self.h:
            raise RgError(-1, "pgen file is closed")
        idx = np.ascontiguousarray(variant_idx, dtype=np.int64)
        rows = np.empty((idx.size, self.n_samples), dtype=np.float64)
        self._check(self.lib.rg_pgen_read_dosage_rows(self.h, idx.size, idx.ctypes.data, rows.ctypes.data, self.n_samples))
        return rows

    def read_hardcalls(self, variant_idx: int) -> np.ndarray:
        """ALT-allele counts 0/1/2 and -3 for missing, as PgenReader::ReadHardcalls(.., allele_idx=1) gives."""
        if not self.h:
            raise RgError(-1, "pgen file is closed")
        out = np.empty(self.n_samples, dtype=np.float64)
        self._check(self.lib.rg_pgen_read_hardcalls(self.h, int(variant_idx), out.ctypes.data))
        return out
