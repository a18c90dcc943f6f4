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

    def read_compressed(self, variant_idx, threads: int = 0, alloc=None):
        """The stored zlib streams of the variants (rg_bgen_read_compressed): (buffer uint8, off int64, clen int32, ulen int32).
        alloc(nbytes) -> uint8 array: where the bytes go (page-locked memory makes the device decoder's copy asynchronous)."""
        idx = np.ascontiguousarray(variant_idx, dtype=np.int64)
        nb = C.c_int64()
        self._check(self.lib.rg_bgen_compressed_bytes(self.h, idx.size, idx.ctypes.data, C.byref(nb)))
        buf = np.zeros(nb.value, dtype=np.uint8) if alloc is None else alloc(nb.value)
        off = np.zeros(idx.size, dtype=np.int64)
        clen = np.zeros(idx.size, dtype=np.int32)
        ulen = np.zeros(idx.size, dtype=np.int32)
        self._check(self.lib.rg_bgen_read_compressed(self.h, idx.size, idx.ctypes.data, buf.ctypes.data, buf.size, off.ctypes.data, clen.ctypes.data,
                                                     ulen.ctypes.data, int(threads)))
        return buf, off, clen, ulen


class RgBgenDevOut(C.Structure):
    _fields_ = [("g16", C.c_void_p), ("ld16", C.c_int64), ("raw", C.c_void_p), ("raw_stride", C.c_int64),
                ("sum_q", C.c_void_p), ("sum_info", C.c_void_p), ("n_obs", C.c_void_p), ("max_q", C.c_void_p),
                ("sum_q_t", C.c_void_p), ("sum_info_t", C.c_void_p), ("n_obs_t", C.c_void_p), ("status", C.c_void_p)]


class BgenDevice:
    """The device path of include/rg_bgen.h (csrc/bgen_inflate.hip): zlib streams inflated one per wavefront, walked into uint16 dosage rows
    and exact integer sums.  Needs a GPU; there is no host fallback behind these calls."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        self.h = C.c_void_p()
        rc = self.lib.rg_bgen_dev_create(C.byref(self.h), int(device))
        if rc != 0:
            raise RgError(rc, "rg_bgen_dev_create failed (no MI355X / HIP device?)")
        self.n = self.n_file = self.P = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.rg_bgen_dev_destroy(self.h)
        self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise RgError(rc, self.lib.rg_bgen_dev_last_error(self.h).decode())

    def set_samples(self, n_file: int, file_idx=None, mask=None):
        fi = None if file_idx is None else np.ascontiguousarray(file_idx, dtype=np.int64)
        n = int(n_file) if fi is None else fi.size
        mk = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        P = 0 if mk is None else mk.shape[0]
        assert mk is None or mk.shape == (P, n)
        self._check(self.lib.rg_bgen_dev_set_samples(self.h, int(n_file), n, None if fi is None else fi.ctypes.data, P, None if mk is None else mk.ctypes.data))
        self.n, self.n_file, self.P = n, int(n_file), P

    def decode(self, comp, off, clen, ulen, ref_first: bool = False, slot: int = 0, fetch_raw: bool = False):
        """Returns a dict: g16 [nvar, n] uint16 (fetched to the host), the sums, status, and with fetch_raw the inflated blocks."""
        nvar = off.size
        o = RgBgenDevOut()
        res = {k: np.zeros(nvar, dtype=np.int64) for k in ("sum_q", "sum_info", "n_obs")}
        res["max_q"] = np.zeros(nvar, dtype=np.int32)
        res["status"] = np.zeros(nvar, dtype=np.int32)
        for k in ("sum_q", "sum_info", "n_obs", "max_q", "status"):
            setattr(o, k, res[k].ctypes.data)
        if self.P:
            for k in ("sum_q_t", "sum_info_t", "n_obs_t"):
                res[k] = np.zeros((nvar, self.P), dtype=np.int64)
                setattr(o, k, res[k].ctypes.data)
        self._check(self.lib.rg_bgen_dev_decode(self.h, int(slot), nvar, comp.ctypes.data, comp.size, off.ctypes.data, clen.ctypes.data, ulen.ctypes.data,
                                                1 if ref_first else 0, C.byref(o)))
        g = np.zeros((nvar, o.ld16), dtype=np.uint16)
        self._check(self.lib.rg_bgen_dev_fetch(self.h, o.g16, g.ctypes.data, g.nbytes))
        res["g16"] = g[:, :self.n]
        res["g16_pad"] = g[:, self.n:]
        res["g16_device_ptr"], res["ld16"] = o.g16, o.ld16
        if fetch_raw:
            raw = np.zeros((nvar, o.raw_stride), dtype=np.uint8)
            self._check(self.lib.rg_bgen_dev_fetch(self.h, o.raw, raw.ctypes.data, raw.nbytes))
            res["raw"] = raw
        return res
