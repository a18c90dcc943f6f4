"""ctypes binding of librg_step1_hip.so -- the product path.

This is the thin host-side mirror of the seam inside regenie's Data::run_step1
(reference src/Data.cpp:95-133): `set_problem` stands where setmem/set_folds do, `l0_blocks` where
the get_G -> residualize_genotypes -> calc_cv_matrices -> ridge_level_0 block loop does
(Data.cpp:636-678), `l1_qt` where ridge_level_1 + output/make_predictions do.

There is NO CPU fallback: if the HIP library cannot be loaded or no GPU is present every call fails
loudly.  Nothing here imports oracle/.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

_LIB = None
RG_MEM_HOST, RG_MEM_DEVICE = 0, 1


class RgProblem(C.Structure):
    _fields_ = [
        ("n_samples", C.c_int64), ("n_file", C.c_int64), ("n_pheno", C.c_int32), ("n_cov", C.c_int32),
        ("cv_folds", C.c_int32), ("n_ridge_l0", C.c_int32), ("ref_first", C.c_int32),
        ("reserved0", C.c_int32), ("n_analyzed", C.c_int64), ("cv_sizes", C.c_void_p),
        ("lambda_", C.c_void_p), ("X", C.c_void_p), ("Y", C.c_void_p), ("mask", C.c_void_p),
        ("ind_in_analysis", C.c_void_p), ("ind_ignore", C.c_void_p), ("neff", C.c_void_p),
        ("n_blocks_total", C.c_int32), ("max_block_size", C.c_int32),
    ]


class RgBtOptions(C.Structure):
    _fields_ = [("niter_max_ridge", C.c_int32), ("niter_max_line_search_ridge", C.c_int32),
                ("niter_max_line_search", C.c_int32), ("family", C.c_int32),
                ("l1_ridge_tol", C.c_double), ("tol", C.c_double), ("beta_out", C.c_void_p), ("fold_cumsum_out", C.c_void_p)]


class RgCoxOptions(C.Structure):
    _fields_ = [("niter_max", C.c_int32), ("niter_max_line_search", C.c_int32), ("niter_max_ridge", C.c_int32),
                ("niter_max_line_search_ridge", C.c_int32), ("numtol_cox", C.c_double), ("l1_ridge_tol", C.c_double),
                ("tau", C.c_void_p)]


class RgTiming(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("ms_prep", "ms_xy", "ms_gram", "ms_assemble", "ms_chol",
                                          "ms_solve", "ms_pred", "ms_l1_gram", "ms_l1_chol",
                                          "ms_l1_pred")] + \
               [("n_gram_launches", C.c_int64), ("n_chol_launches", C.c_int64)] + \
               [("ms_wgram", C.c_double), ("ms_irls_solve", C.c_double), ("ms_irls_stream", C.c_double),
                ("n_wgram", C.c_int64), ("n_irls_rounds", C.c_int64), ("wgram_positions", C.c_int64), ("n_wgram_approx_rounds", C.c_int64), ("n_irls_passes", C.c_int64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


EXPORTS = ["rg_create", "rg_destroy", "rg_last_error", "rg_set_problem", "rg_set_l0_workspace", "rg_w_rows", "rg_w_bytes",
           "rg_set_w_buffer", "rg_set_block_range", "rg_w_device_ptr", "rg_l0_blocks", "rg_l0_blocks_f64", "rg_sync", "rg_l0_get_w",
           "rg_l0_set_w", "rg_l1_qt", "rg_l1_qt_loocv", "rg_l1_bt", "rg_l1_cox", "rg_set_collective", "rg_set_l1_view", "rg_set_loco_output", "rg_enable_timing", "rg_get_timing", "rg_k_gram_i8", "rg_k_gram_fp4",
           "rg_k_chol_solve", "rg_k_dgemm_nt", "rg_k_mfma_peak",
           # one node, several GPUs: level-0 hand-off over RCCL / peer copies; streamed ingest helpers (used by the C++ driver)
           "rg_group_create", "rg_group_destroy", "rg_l0_finish", "rg_group_prepare", "rg_group_abort", "rg_l0_batch_blocks", "rg_host_alloc", "rg_host_free", "rg_host_register", "rg_host_unregister",
           "rg_ingest_fence", "rg_stage_alloc", "rg_stage_copy", "rg_stage_free", "rg_stage_fits",
           # include/rg_pgen.h (host-side .pgen hardcall input; wrapped by regenie_amd/pgen.py)
           "rg_pgen_open", "rg_pgen_close", "rg_pgen_last_error", "rg_pgen_info", "rg_pgen_read_bed_rows",
           "rg_pgen_read_hardcalls", "rg_pgen_set_threads", "rg_pgen_read_dosages", "rg_pgen_read_dosage_rows",
           # include/rg_bgen.h (host-side BGEN v1.2 input; wrapped by regenie_amd/bgen.py)
           "rg_bgen_open", "rg_bgen_close", "rg_bgen_last_error", "rg_bgen_info", "rg_bgen_sample_id", "rg_bgen_variant",
           "rg_bgen_set_threads", "rg_bgen_read_dosages", "rg_bgen_read_dosages_info", "rg_bgen_block_bytes", "rg_bgen_read_blocks",
           # its device path (csrc/bgen_inflate.hip): the stored zlib streams, inflated and walked on the GPU
           "rg_bgen_compressed_bytes", "rg_bgen_read_compressed", "rg_bgen_dev_create", "rg_bgen_dev_destroy", "rg_bgen_dev_last_error",
           "rg_bgen_dev_set_samples", "rg_bgen_dev_decode", "rg_bgen_dev_fetch",
           # include/rg_step2.h (Step-2 QT score test; wrapped by regenie_amd/step2.py)
           "rg_s2_create", "rg_s2_destroy", "rg_s2_last_error", "rg_s2_set_null", "rg_s2_qt_block", "rg_s2_qt_block_packed", "rg_s2_qt_block_int", "rg_s2_set_sparse_rule", "rg_s2_set_columns", "rg_s2_contract_packed", "rg_s2_contract_int", "rg_s2_bt_set_null", "rg_s2_bt_score_packed", "rg_s2_bt_score_int", "rg_s2_bt_correct",
           "rg_s2_last_kernel_ms"]


def lib_path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "librg_step1_hip.so")


def load_library() -> C.CDLL:
    """Loads the in-tree HIP library; raises if it is missing (no silent fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError("librg_step1_hip.so is not built (%s): run `python -m regenie_amd.build` "
                           "or __graft_entry__.build()" % path)
    lib = C.CDLL(path)
    lib.rg_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
    lib.rg_destroy.argtypes = [C.c_void_p]
    lib.rg_destroy.restype = None
    lib.rg_last_error.argtypes = [C.c_void_p]
    lib.rg_last_error.restype = C.c_char_p
    lib.rg_set_problem.argtypes = [C.c_void_p, C.POINTER(RgProblem)]
    lib.rg_w_rows.argtypes = [C.c_void_p]
    lib.rg_w_rows.restype = C.c_int64
    lib.rg_w_bytes.argtypes = [C.c_void_p]
    lib.rg_w_bytes.restype = C.c_int64
    lib.rg_set_w_buffer.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.rg_w_device_ptr.argtypes = [C.c_void_p]
    lib.rg_w_device_ptr.restype = C.c_void_p
    lib.rg_l0_blocks.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
    lib.rg_l0_blocks_f64.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
    lib.rg_sync.argtypes = [C.c_void_p]
    lib.rg_l0_get_w.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.rg_l0_set_w.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.rg_l1_qt.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_void_p]
    lib.rg_l1_qt_loocv.argtypes = lib.rg_l1_qt.argtypes
    lib.rg_l1_bt.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rg_l1_cox.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rg_set_collective.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.rg_set_l1_view.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
    lib.rg_set_loco_output.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
    lib.rg_set_block_range.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    lib.rg_group_create.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p), C.c_int]
    lib.rg_group_destroy.argtypes = [C.c_void_p]
    lib.rg_group_destroy.restype = None
    lib.rg_l0_finish.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.rg_l0_batch_blocks.argtypes = [C.c_void_p]
    lib.rg_l0_batch_blocks.restype = C.c_int32
    lib.rg_host_alloc.argtypes = [C.c_int64]
    lib.rg_host_alloc.restype = C.c_void_p
    lib.rg_host_free.argtypes = [C.c_void_p]
    lib.rg_host_free.restype = None
    lib.rg_ingest_fence.argtypes = [C.c_void_p]
    lib.rg_stage_alloc.argtypes = [C.c_void_p, C.c_int64, C.c_double]
    lib.rg_stage_alloc.restype = C.c_void_p
    lib.rg_stage_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    lib.rg_stage_free.argtypes = [C.c_void_p, C.c_void_p]
    lib.rg_stage_free.restype = None
    lib.rg_stage_fits.argtypes = [C.c_void_p, C.c_int64]
    lib.rg_enable_timing.argtypes = [C.c_void_p, C.c_int]
    lib.rg_get_timing.argtypes = [C.c_void_p, C.POINTER(RgTiming)]
    lib.rg_k_gram_i8.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64,
                                 C.c_int, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_int64]
    lib.rg_k_gram_fp4.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                  C.c_int64, C.c_void_p, C.c_int64]
    lib.rg_k_chol_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.rg_k_dgemm_nt.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                  C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_int64]
    lib.rg_k_mfma_peak.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double)]
    lib.rg_pgen_open.argtypes = [C.POINTER(C.c_void_p), C.c_char_p]
    lib.rg_pgen_close.argtypes = [C.c_void_p]
    lib.rg_pgen_close.restype = None
    lib.rg_pgen_last_error.argtypes = [C.c_void_p]
    lib.rg_pgen_last_error.restype = C.c_char_p
    lib.rg_pgen_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                 C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.rg_pgen_read_dosages.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    lib.rg_pgen_read_dosage_rows.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64]
    lib.rg_s2_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int32, C.c_int32]
    lib.rg_s2_destroy.argtypes = [C.c_void_p]
    lib.rg_s2_destroy.restype = None
    lib.rg_s2_last_error.argtypes = [C.c_void_p]
    lib.rg_s2_last_error.restype = C.c_char_p
    lib.rg_s2_set_null.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rg_s2_qt_block.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_double, C.c_void_p]
    lib.rg_s2_qt_block_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_void_p]
    lib.rg_s2_qt_block_int.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_void_p]
    lib.rg_s2_set_sparse_rule.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_int32]
    lib.rg_s2_set_columns.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
    lib.rg_s2_contract_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.rg_s2_contract_int.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.rg_s2_bt_set_null.argtypes = [C.c_void_p, C.c_void_p]
    lib.rg_s2_bt_score_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_void_p]
    lib.rg_s2_bt_score_int.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_void_p]
    lib.rg_s2_bt_correct.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.rg_s2_last_kernel_ms.argtypes = [C.c_void_p]
    lib.rg_s2_last_kernel_ms.restype = C.c_double
    lib.rg_bgen_open.argtypes = [C.POINTER(C.c_void_p), C.c_char_p]
    lib.rg_bgen_close.argtypes = [C.c_void_p]
    lib.rg_bgen_close.restype = None
    lib.rg_bgen_last_error.argtypes = [C.c_void_p]
    lib.rg_bgen_last_error.restype = C.c_char_p
    lib.rg_bgen_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.rg_bgen_sample_id.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_char_p)]
    lib.rg_bgen_variant.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.POINTER(C.c_char_p),
                                    C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_int64)]
    lib.rg_bgen_set_threads.argtypes = [C.c_void_p, C.c_int32]
    lib.rg_bgen_read_dosages.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
    lib.rg_bgen_read_dosages_info.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64]
    lib.rg_bgen_block_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    lib.rg_bgen_read_blocks.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
    lib.rg_bgen_compressed_bytes.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int64)]
    lib.rg_bgen_read_compressed.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    lib.rg_bgen_dev_create.argtypes = [C.POINTER(C.c_void_p), C.c_int32]
    lib.rg_bgen_dev_destroy.argtypes = [C.c_void_p]
    lib.rg_bgen_dev_destroy.restype = None
    lib.rg_bgen_dev_last_error.argtypes = [C.c_void_p]
    lib.rg_bgen_dev_last_error.restype = C.c_char_p
    lib.rg_bgen_dev_set_samples.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]
    lib.rg_bgen_dev_decode.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.rg_bgen_dev_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    lib.rg_pgen_read_bed_rows.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64]
    lib.rg_pgen_read_hardcalls.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    lib.rg_pgen_set_threads.argtypes = [C.c_void_p, C.c_int32]
    _LIB = lib
    return lib


def mfma_peak(kind: int, iters: int = 2000) -> float:
    """Measured register-only MFMA rate (kind 0: fp64 TFLOP/s, kind 1: i8 TOP/s)."""
    out = C.c_double(0.0)
    rc = load_library().rg_k_mfma_peak(kind, iters, C.byref(out))
    if rc != 0:
        raise RgError(rc, "rg_k_mfma_peak failed")
    return out.value


class RgError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("rg error %d: %s" % (code, msg))
        self.code = code


class Step1Engine:
    """One context per process / GPU."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.rg_create(C.byref(h), device, C.c_void_p(stream) if stream else None)
        if rc != 0 or not h:
            raise RgError(rc, "rg_create failed (no MI355X / HIP device visible?)")
        self.h = h
        self._keep = []
        self.N = self.P = self.R0 = self.B = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.rg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            raise RgError(rc, self.lib.rg_last_error(self.h).decode())

    def set_problem(self, *, X, Y, mask, ind_in_analysis, cv_sizes, lam, neff, n_file,
                    n_blocks_total, max_block_size, ind_ignore=None, ref_first=False):
        X = np.asfortranarray(X, dtype=np.float64)
        Y = np.asfortranarray(Y, dtype=np.float64)
        mask = np.asfortranarray(mask, dtype=np.uint8)
        ain = np.ascontiguousarray(ind_in_analysis, dtype=np.uint8)
        # cv_sizes=None selects leave-one-out CV (cv_folds = 0 in the ABI)
        cvs = np.ascontiguousarray(cv_sizes if cv_sizes is not None else [], dtype=np.int32)
        lam = np.ascontiguousarray(lam, dtype=np.float64)
        neff = np.ascontiguousarray(neff, dtype=np.float64)
        ign = None if ind_ignore is None else np.ascontiguousarray(ind_ignore, dtype=np.uint8)
        N, P = Y.shape
        p = RgProblem()
        p.n_samples, p.n_file, p.n_pheno, p.n_cov = N, int(n_file), P, X.shape[1]
        p.cv_folds, p.n_ridge_l0, p.ref_first = cvs.size, lam.size, int(bool(ref_first))
        p.n_analyzed = int(ain.sum())
        p.cv_sizes, p.lambda_ = (cvs.ctypes.data if cvs.size else None), lam.ctypes.data
        p.X, p.Y, p.mask = X.ctypes.data, Y.ctypes.data, mask.ctypes.data
        p.ind_in_analysis = ain.ctypes.data
        p.ind_ignore = ign.ctypes.data if ign is not None else None
        p.neff = neff.ctypes.data
        p.n_blocks_total, p.max_block_size = int(n_blocks_total), int(max_block_size)
        self._check(self.lib.rg_set_problem(self.h, C.byref(p)))
        self.N, self.P, self.R0, self.B = N, P, lam.size, int(n_blocks_total)
        self.cv_folds = int(cvs.size)
        self.Pv = P                       # phenotypes of the current level-1 view
        self.n_file = int(n_file)

    @property
    def w_rows(self) -> int:
        return self.lib.rg_w_rows(self.h)

    @property
    def w_bytes(self) -> int:
        return self.lib.rg_w_bytes(self.h)

    def set_w_buffer(self, dev_ptr: int, nbytes: int):
        self._check(self.lib.rg_set_w_buffer(self.h, C.c_void_p(dev_ptr), nbytes))

    def set_block_range(self, first_block: int, n_blocks: int):
        """W then holds the predictor rows of blocks [first_block, first_block + n_blocks) only (rg_set_block_range)."""
        self._check(self.lib.rg_set_block_range(self.h, int(first_block), int(n_blocks)))

    def l0_blocks_host(self, block_ids: Sequence[int], rows: List[np.ndarray]):
        """rows[b]: (bs_b, ceil(N_file/4)) uint8 C-contiguous packed .bed rows (host memory)."""
        nb = len(block_ids)
        ids = np.ascontiguousarray(block_ids, dtype=np.int32)
        bs = np.ascontiguousarray([r.shape[0] for r in rows], dtype=np.int32)
        rows = [np.ascontiguousarray(r, dtype=np.uint8) for r in rows]
        stride = rows[0].shape[1]
        assert all(r.shape[1] == stride for r in rows)
        ptrs = (C.c_void_p * nb)(*[r.ctypes.data for r in rows])
        self._check(self.lib.rg_l0_blocks(self.h, nb, ids.ctypes.data, bs.ctypes.data, ptrs, stride, RG_MEM_HOST))

    def l0_blocks_f64_host(self, block_ids: Sequence[int], rows: List[np.ndarray]):
        """Level 0 on non-integer genotypes: rows[b] is (bs_b, N_file) float64, C-contiguous, file sample order, ALT dosages
        in [0, 2] with -3 for missing (what PgenFile.read_dosages returns).  K-fold or leave-one-out CV as the problem was set up."""
        nb = len(block_ids)
        ids = np.ascontiguousarray(block_ids, dtype=np.int32)
        bs = np.ascontiguousarray([r.shape[0] for r in rows], dtype=np.int32)
        rows = [np.ascontiguousarray(r, dtype=np.float64) for r in rows]
        stride = rows[0].shape[1]
        assert all(r.shape[1] == stride for r in rows)
        ptrs = (C.c_void_p * nb)(*[r.ctypes.data for r in rows])
        self._check(self.lib.rg_l0_blocks_f64(self.h, nb, ids.ctypes.data, bs.ctypes.data, ptrs, stride, RG_MEM_HOST))

    def l0_blocks_device(self, block_ids: Sequence[int], bs: Sequence[int], dev_ptrs: Sequence[int], row_stride: int):
        nb = len(block_ids)
        ids = np.ascontiguousarray(block_ids, dtype=np.int32)
        bsa = np.ascontiguousarray(bs, dtype=np.int32)
        ptrs = (C.c_void_p * nb)(*[int(p) for p in dev_ptrs])
        self._check(self.lib.rg_l0_blocks(self.h, nb, ids.ctypes.data, bsa.ctypes.data, ptrs, int(row_stride), RG_MEM_DEVICE))

    def sync(self):
        self._check(self.lib.rg_sync(self.h))

    def get_w(self, block_id: int, pheno: int) -> np.ndarray:
        out = np.empty((self.N, self.R0), dtype=np.float64, order="F")
        self._check(self.lib.rg_l0_get_w(self.h, block_id, pheno, out.ctypes.data))
        return out

    def set_w(self, block_id: int, pheno: int, w: np.ndarray):
        w = np.asfortranarray(w, dtype=np.float64)
        assert w.shape == (self.N, self.R0)
        self._check(self.lib.rg_l0_set_w(self.h, block_id, pheno, w.ctypes.data))

    def l1_qt(self, tau: np.ndarray, cols_per_chr: Sequence[int]):
        """tau: (P, R1) scaled ridge values.  Returns (cumsum [P,5,R1], best [P], pred [P][N,nchr])."""
        tau = np.ascontiguousarray(tau, dtype=np.float64)
        P, R1 = tau.shape
        assert P == self.Pv
        cpc = np.ascontiguousarray(cols_per_chr, dtype=np.int32)
        nchr = cpc.size
        cs = np.zeros((P, 5, R1))
        best = np.zeros(P, dtype=np.int32)
        pred = self._host_out((P, self._pred_rows(nchr), self.N))
        self._check(self.lib.rg_l1_qt(self.h, R1, tau.ctypes.data, nchr, cpc.ctypes.data, cs.ctypes.data,
                                      best.ctypes.data, pred.ctypes.data))
        return cs, best, [pred[p].T for p in range(P)]

    def _host_out(self, shape):
        """Reusable host buffer for the per-chromosome predictions (every entry is overwritten by the call): page-locked
        when torch can provide it, so that the device -> host copy (0.9 GB at 500k samples x 10 phenotypes) runs at PCIe
        speed instead of faulting fresh pages in.  The views returned by l1_qt stay valid until its next call."""
        bufs = self.__dict__.setdefault("_hostbufs", {})
        key = tuple(int(x) for x in shape)
        if key not in bufs:
            arr = None
            try:
                import torch
                t = torch.empty(key, dtype=torch.float64, pin_memory=bool(torch.cuda.is_available()))
                arr = (t, t.numpy())
            except Exception:  # noqa: BLE001 - no torch / no pinned memory: plain pages, touched once
                arr = (None, np.zeros(key))
            bufs[key] = arr
        return bufs[key][1]

    def l1_qt_loocv(self, tau: np.ndarray, cols_per_chr: Sequence[int]):
        """Leave-one-out level 1 (problem set up with cv_sizes=None).  Same returns as l1_qt."""
        tau = np.ascontiguousarray(tau, dtype=np.float64)
        P, R1 = tau.shape
        assert P == self.Pv
        cpc = np.ascontiguousarray(cols_per_chr, dtype=np.int32)
        nchr = cpc.size
        cs = np.zeros((P, 5, R1))
        best = np.zeros(P, dtype=np.int32)
        pred = np.zeros((P, self._pred_rows(nchr), self.N))
        self._check(self.lib.rg_l1_qt_loocv(self.h, R1, tau.ctypes.data, nchr, cpc.ctypes.data, cs.ctypes.data,
                                            best.ctypes.data, pred.ctypes.data))
        return cs, best, [pred[p].T.copy() for p in range(P)]

    def l1_bt(self, tau: np.ndarray, yraw: np.ndarray, offset: np.ndarray, cols_per_chr: Sequence[int],
              niter_max_ridge: int = 100, niter_max_line_search_ridge: int = 100,
              niter_max_line_search: int = 25, l1_ridge_tol: float = 1e-4, tol: float = 1e-8, family: int = 0, fold_detail: bool = False):
        """Logistic (family 0, --bt) or Poisson (family 1, --ct) ridge level 1, K-fold or LOOCV as the problem was set up.
        Returns (cumsum [P,6,R1], converged [P] bool, best [P], pred [P][N,nchr]); with fold_detail (K-fold) also the fold models'
        coefficients [P,K,R1,L] and each fold's own held-out sums [P,K,6,R1] (rg_bt_options.beta_out / fold_cumsum_out)."""
        tau = np.ascontiguousarray(tau, dtype=np.float64)
        P, R1 = tau.shape
        assert P == self.Pv
        yraw = np.asfortranarray(yraw, dtype=np.float64)
        offset = np.asfortranarray(offset, dtype=np.float64)
        assert yraw.shape == (self.N, P) and offset.shape == (self.N, P)
        cpc = np.ascontiguousarray(cols_per_chr, dtype=np.int32)
        nchr = cpc.size
        o = RgBtOptions(niter_max_ridge, niter_max_line_search_ridge, niter_max_line_search, family, l1_ridge_tol, tol, None, None)
        if fold_detail:
            K, L = int(self.cv_folds), self.B * self.R0
            assert K > 0, "fold_detail needs a K-fold problem"
            betas = np.zeros((P, K, R1, L))
            fcs = np.zeros((P, K, 6, R1))
            o.beta_out, o.fold_cumsum_out = betas.ctypes.data, fcs.ctypes.data
        cs = np.zeros((P, 6, R1))
        conv = np.zeros(P, dtype=np.int32)
        best = np.zeros(P, dtype=np.int32)
        pred = np.zeros((P, self._pred_rows(nchr), self.N))
        self._check(self.lib.rg_l1_bt(self.h, R1, tau.ctypes.data, yraw.ctypes.data, offset.ctypes.data,
                                      C.byref(o), nchr, cpc.ctypes.data, cs.ctypes.data, conv.ctypes.data,
                                      best.ctypes.data, pred.ctypes.data))
        if fold_detail:
            return cs, conv.astype(bool), best, [pred[p].T.copy() for p in range(P)], betas, fcs
        return cs, conv.astype(bool), best, [pred[p].T.copy() for p in range(P)]

    def l1_cox(self, pheno: int, time: np.ndarray, event: np.ndarray, offset: np.ndarray, cols_per_chr: Sequence[int], n_ridge_l1: int = 5,
               niter_max_ridge: int = 100, niter_max_line_search_ridge: int = 100, l1_ridge_tol: float = 1e-4, tau_in: Optional[np.ndarray] = None):
        """Cox ridge level 1 of one time-to-event trait (--t2e; K-fold).  Returns (tau [R1], deviance [R1], converged, best, pred [N,nchr]).
        tau_in: the caller's penalties (--t2e-l1-pi6) instead of the path from the score at beta = 0."""
        time = np.ascontiguousarray(time, dtype=np.float64)
        event = np.ascontiguousarray(event, dtype=np.float64)
        offset = np.ascontiguousarray(offset, dtype=np.float64)
        assert time.shape == event.shape == offset.shape == (self.N,)
        cpc = np.ascontiguousarray(cols_per_chr, dtype=np.int32)
        nchr = cpc.size
        if tau_in is not None:
            tau_in = np.ascontiguousarray(tau_in, dtype=np.float64)
            assert tau_in.shape == (n_ridge_l1,)
        o = RgCoxOptions(50, 25, niter_max_ridge, niter_max_line_search_ridge, 2.5e-4, l1_ridge_tol, tau_in.ctypes.data if tau_in is not None else None)
        tau = np.zeros(n_ridge_l1)
        dev = np.zeros(n_ridge_l1)
        conv = C.c_int32(0)
        best = C.c_int32(0)
        pred = np.zeros((self._pred_rows(nchr), self.N))
        self._check(self.lib.rg_l1_cox(self.h, int(pheno), int(n_ridge_l1), time.ctypes.data, event.ctypes.data, offset.ctypes.data, C.byref(o),
                                       nchr, cpc.ctypes.data, tau.ctypes.data, dev.ctypes.data, C.byref(conv), C.byref(best), pred.ctypes.data))
        return tau, dev, bool(conv.value), int(best.value), pred.T.copy()

    def set_loco_output(self, chroms: Optional[Sequence[int]], nchrom: int = 23):
        """LOCO output mode: the l1_* calls then return, per phenotype, the (N, nchrom) LOCO predictions (column c-1 leaves
        chromosome c out) assembled on the device instead of the (N, nchr) per-chromosome predictions.  chroms[k] is the
        chromosome (1-based) of the k-th entry of cols_per_chr; None switches back."""
        if chroms is None:
            self._check(self.lib.rg_set_loco_output(self.h, 0, None, 0))
            self._loco_rows = 0
            return
        ids = np.ascontiguousarray(chroms, dtype=np.int32)
        self._check(self.lib.rg_set_loco_output(self.h, int(nchrom), ids.ctypes.data, ids.size))
        self._loco_rows = int(nchrom)

    def _pred_rows(self, nchr: int) -> int:
        return getattr(self, "_loco_rows", 0) or nchr

    def set_l1_view(self, w_dev_ptr, pheno_begin: int, pheno_count: int):
        """Phenotype-sharded level 1: the following l1_* calls work on phenotypes [begin, begin+count) and read the
        predictors from the device buffer w_dev_ptr laid out [L][count][w_rows]; (None, 0, P) restores the default."""
        self._check(self.lib.rg_set_l1_view(self.h, C.c_void_p(w_dev_ptr) if w_dev_ptr else None, pheno_begin, pheno_count))
        self.Pv = pheno_count

    def set_collective(self, world: int, rank: int, allreduce=None):
        """Shares level 1 among `world` ranks.  allreduce(dev_ptr: int, n_doubles: int) must sum the device
        buffer in place over all ranks (e.g. torch.distributed.all_reduce on a tensor view of it)."""
        cb_t = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64)

        def _cb(_user, ptr, n):
            try:
                allreduce(int(ptr), int(n))
                return 0
            except Exception as e:  # noqa: BLE001 - reported through the C return code
                import sys
                print("all-reduce callback failed:", e, file=sys.stderr)
                return 1
        self._coll_cb = cb_t(_cb) if allreduce is not None else None
        self._check(self.lib.rg_set_collective(self.h, world, rank,
                                               C.cast(self._coll_cb, C.c_void_p) if self._coll_cb else None, None))

    def enable_timing(self, on: bool = True):
        self._check(self.lib.rg_enable_timing(self.h, int(on)))

    def timing(self) -> dict:
        t = RgTiming()
        self._check(self.lib.rg_get_timing(self.h, C.byref(t)))
        return t.as_dict()


def loco_from_predictions(pred: np.ndarray, chroms: Sequence[int], nchrom: int = 23) -> np.ndarray:
    """write_predictions' LOCO assembly (reference src/Data.cpp:1846-1858): LOCO[:,c] = rowsum - pred[:,c];
    chromosomes without blocks get the full sum.  pred: (N, nchr); works on chromosome-major rows so that a
    transposed view of the library's [nchr][N] output is processed without strided passes."""
    pt = pred.T                                     # (nchr, N); contiguous when pred is a transposed view
    tot = pt.sum(axis=0)
    out = np.empty((nchrom, pred.shape[0]))
    out[:] = tot
    for ci, c in enumerate(chroms):
        out[c - 1] -= pt[ci]
    return out.T
