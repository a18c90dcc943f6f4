"""Step-2 quantitative-trait score test (`regenie --step 2 --qt`, dense genotypes): ctypes wrapper over include/rg_step2.h
(regenie_amd/csrc/step2_qt.hip).  Mirrors the reference's per-chromosome / per-block split: `set_null` is what
Data::compute_res leaves behind (Data.cpp:2386-2400), `score_block` is compute_tests_mt over one block
(Data.cpp:2476-2555 -> Geno.cpp:3242-3260 -> Step2_Models.cpp:343-468).  No CPU path: without the HIP library or a
GPU the constructor raises."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .engine import RgError, load_library


class _QtOut(C.Structure):
    _fields_ = [("stats", C.c_void_p), ("bhat", C.c_void_p), ("scale_fac", C.c_void_p), ("mean", C.c_void_p),
                ("n_obs", C.c_void_p), ("ignored", C.c_void_p), ("total_p", C.c_void_p), ("n_obs_p", C.c_void_p)]


class _ContractOut(C.Structure):
    _fields_ = [("sums", C.c_void_p), ("sq", C.c_void_p), ("counts", C.c_void_p), ("vstat", C.c_void_p)]


class _BtNull(C.Structure):
    _fields_ = [("family", C.c_int32), ("niter_max", C.c_int32), ("X", C.c_void_p), ("y", C.c_void_p), ("mask", C.c_void_p),
                ("fitted", C.c_void_p), ("firth_offset", C.c_void_p), ("pass_", C.c_void_p)]


class _BtOut(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("stats", "bhat", "denum", "test_ignored", "mean", "ignored", "sparse", "counts", "vstat",
                                           "total_p", "n_obs_p")]


class _BtCorr(C.Structure):
    _fields_ = [("beta", C.c_double), ("se", C.c_double), ("chisq", C.c_double), ("logp", C.c_double), ("fail", C.c_int32),
                ("reserved", C.c_int32)]


BT_FIRTH_APPROX, BT_SPA = 1, 2


class Step2QT:
    NUMTOL = 1e-6     # params.numtol, Regenie.hpp:220

    def __init__(self, n: int, n_cov: int, n_pheno: int, device: int = 0):
        self.lib = load_library()
        self.h = C.c_void_p()
        self.n, self.C, self.P = int(n), int(n_cov), int(n_pheno)
        rc = self.lib.rg_s2_create(C.byref(self.h), int(device), self.n, self.C, self.P)
        if rc != 0:
            msg = self.lib.rg_s2_last_error(self.h).decode() if self.h else "rg_s2_create failed"
            self.close()
            raise RgError(rc, msg)

    def close(self) -> None:
        if getattr(self, "h", None):
            self.lib.rg_s2_destroy(self.h)
        self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        self.close()

    def _check(self, rc: int) -> None:
        if rc != 0:
            raise RgError(rc, self.lib.rg_s2_last_error(self.h).decode())

    def set_null(self, X: np.ndarray, yres: np.ndarray, mask: np.ndarray, scf_sv: np.ndarray) -> None:
        """X [C][n] orthonormal covariate basis (new_cov^T), yres [P][n] masked + scaled LOCO residuals (res^T),
        mask [P][n] (masked_indivs^T), scf_sv [P] = scale_Y * p_sd_yres."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        yres = np.ascontiguousarray(yres, dtype=np.float64)
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        scf_sv = np.ascontiguousarray(scf_sv, dtype=np.float64)
        if X.shape != (self.C, self.n) or yres.shape != (self.P, self.n) or mask.shape != (self.P, self.n) \
                or scf_sv.shape != (self.P,):
            raise ValueError("set_null: expected X %s, yres/mask %s, scf_sv (%d,)" % ((self.C, self.n), (self.P, self.n), self.P))
        self._check(self.lib.rg_s2_set_null(self.h, X.ctypes.data, yres.ctypes.data, mask.ctypes.data, scf_sv.ctypes.data))

    def set_sparse_rule(self, n_samples: int, prop_zero_thr: float = 0.5, zero_count_rule: bool = False) -> None:
        """check_sparse_G's constants (Geno.cpp:3165-3177): params.n_samples (kept samples of the file) and --prop-zero-thr;
        zero_count_rule = the .pgen form (observed zeros >= n_samples * thr)."""
        self._check(self.lib.rg_s2_set_sparse_rule(self.h, int(n_samples), float(prop_zero_thr), 1 if zero_count_rule else 0))

    def score_block(self, G, numtol: float = NUMTOL) -> dict:
        """G: numpy [bs][n] float64 (host), or a CUDA torch tensor [bs][n] float64 (read in place).  Missing = NaN or < 0."""
        on_device = 0
        if isinstance(G, np.ndarray):
            G = np.ascontiguousarray(G, dtype=np.float64)
            bs, ld, ptr = G.shape[0], G.shape[1], G.ctypes.data
            if G.shape[1] != self.n:
                raise ValueError("score_block: G must be [bs][n]")
        else:   # torch tensor on the device
            if not (G.is_cuda and G.dtype.is_floating_point and G.element_size() == 8 and G.dim() == 2 and G.stride(1) == 1):
                raise ValueError("score_block: device G must be a 2-d float64 CUDA tensor with unit sample stride")
            bs, ld, ptr, on_device = G.shape[0], G.stride(0), G.data_ptr(), 1
            if G.shape[1] != self.n:
                raise ValueError("score_block: G must be [bs][n]")
            import torch
            torch.cuda.current_stream(G.device).synchronize()   # the library runs on its own stream
        res = {"stats": np.empty((bs, self.P)), "bhat": np.empty((bs, self.P)), "scale_fac": np.empty(bs),
               "mean": np.empty(bs), "n_obs": np.empty(bs, np.int32), "ignored": np.empty(bs, np.int32)}
        out = _QtOut(*[res[k].ctypes.data for k in ("stats", "bhat", "scale_fac", "mean", "n_obs", "ignored")])
        self._check(self.lib.rg_s2_qt_block(self.h, ptr, ld, bs, on_device, float(numtol), C.byref(out)))
        return self._finish(res)

    def _finish(self, res: dict) -> dict:
        res["kernel_ms"] = self.lib.rg_s2_last_kernel_ms(self.h)
        with np.errstate(invalid="ignore", divide="ignore"):
            res["se"] = res["bhat"] / res["stats"]          # Step2_Models.cpp:440
            res["chisq"] = res["stats"] ** 2                # Step2_Models.cpp:443
        return res

    def score_block_packed(self, rows, flip: bool = False, numtol: float = NUMTOL) -> dict:
        """Hard calls as they lie in a .bed file: rows [bs][>= ceil(n/4)] uint8 (numpy, or a CUDA torch tensor read in place),
        2 bits per analysed sample (00 -> 2, 01 -> missing, 10 -> 1, 11 -> 0 copies of the counted allele); flip = --ref-first.
        Follows the reference's per-variant choice between the sparse and the dense branch of compute_score_qt (set_sparse_rule);
        extra outputs total_p / n_obs_p [bs][P]: allele count and number of the samples observed for the variant and the trait."""
        on_device = 0
        if isinstance(rows, np.ndarray):
            rows = np.ascontiguousarray(rows, dtype=np.uint8)
            bs, ld, ptr = rows.shape[0], rows.shape[1], rows.ctypes.data
        else:
            if not (rows.is_cuda and rows.element_size() == 1 and rows.dim() == 2 and rows.stride(1) == 1):
                raise ValueError("score_block_packed: device rows must be a 2-d uint8 CUDA tensor with unit byte stride")
            bs, ld, ptr, on_device = rows.shape[0], rows.stride(0), rows.data_ptr(), 1
            import torch
            torch.cuda.current_stream(rows.device).synchronize()
        if rows.shape[1] < (self.n + 3) // 4:
            raise ValueError("score_block_packed: rows must hold ceil(n / 4) bytes")
        res = {"stats": np.empty((bs, self.P)), "bhat": np.empty((bs, self.P)), "scale_fac": np.empty(bs),
               "mean": np.empty(bs), "n_obs": np.empty(bs, np.int32), "ignored": np.empty(bs, np.int32),
               "total_p": np.empty((bs, self.P)), "n_obs_p": np.empty((bs, self.P), np.int32)}
        out = _QtOut(*[res[k].ctypes.data for k in ("stats", "bhat", "scale_fac", "mean", "n_obs", "ignored", "total_p", "n_obs_p")])
        self._check(self.lib.rg_s2_qt_block_packed(self.h, ptr, ld, bs, on_device, 1 if flip else 0, float(numtol), C.byref(out)))
        return self._finish(res)

    def score_block_int(self, G: np.ndarray, scale: int, numtol: float = NUMTOL) -> dict:
        """Integer dosages: G [bs][n] uint16 in units of 1 / scale (255 for 8-bit .bgen probabilities, 16384 for .pgen), 0xFFFF = missing."""
        G = np.ascontiguousarray(G, dtype=np.uint16)
        if G.ndim != 2 or G.shape[1] != self.n:
            raise ValueError("score_block_int: G must be [bs][n]")
        bs = G.shape[0]
        res = {"stats": np.empty((bs, self.P)), "bhat": np.empty((bs, self.P)), "scale_fac": np.empty(bs),
               "mean": np.empty(bs), "n_obs": np.empty(bs, np.int32), "ignored": np.empty(bs, np.int32)}
        out = _QtOut(*[res[k].ctypes.data for k in ("stats", "bhat", "scale_fac", "mean", "n_obs", "ignored")])
        self._check(self.lib.rg_s2_qt_block_int(self.h, G.ctypes.data, G.shape[1], bs, 0, int(scale), float(numtol), C.byref(out)))
        return self._finish(res)

    # ---- the contraction primitive (rg_s2_set_columns / rg_s2_contract_packed) ------------------------------------------------
    def set_columns(self, cols: np.ndarray, n_sq: int = 0) -> None:
        """cols [n_col][n] float64: the fixed columns the hard-call rows are contracted with; the first n_sq also against g^2."""
        cols = np.ascontiguousarray(cols, dtype=np.float64)
        if cols.ndim != 2 or cols.shape[1] != self.n:
            raise ValueError("set_columns: cols must be [n_col][n]")
        self._ncol, self._nsq = cols.shape[0], int(n_sq)
        self._check(self.lib.rg_s2_set_columns(self.h, cols.shape[0], cols.ctypes.data, int(n_sq)))

    def contract_packed(self, rows, flip: bool = False) -> dict:
        """-> sums [bs][2][n_col] (allele count . column, missing indicator . column), sq [bs][n_sq], counts [bs][4] (ones, twos, missing, 0)."""
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        bs = rows.shape[0]
        res = {"sums": np.zeros((bs, 2, self._ncol)), "sq": np.zeros((bs, self._nsq)), "counts": np.zeros((bs, 4), np.int32)}
        out = _ContractOut(res["sums"].ctypes.data, res["sq"].ctypes.data if self._nsq else None, res["counts"].ctypes.data, None)
        self._check(self.lib.rg_s2_contract_packed(self.h, rows.ctypes.data, rows.shape[1], bs, 0, 1 if flip else 0, C.byref(out)))
        res["kernel_ms"] = self.lib.rg_s2_last_kernel_ms(self.h)
        return res

    def contract_int(self, G: np.ndarray, scale: int) -> dict:
        """The same sums for integer dosages (uint16, units of 1 / scale, 0xFFFF = missing), in genotype units; vstat [bs][4] = sum,
        sum of squares (integer units), observed count, observed non-zero count."""
        G = np.ascontiguousarray(G, dtype=np.uint16)
        bs = G.shape[0]
        res = {"sums": np.zeros((bs, 2, self._ncol)), "sq": np.zeros((bs, self._nsq)), "vstat": np.zeros((bs, 4))}
        out = _ContractOut(res["sums"].ctypes.data, res["sq"].ctypes.data if self._nsq else None, None, res["vstat"].ctypes.data)
        self._check(self.lib.rg_s2_contract_int(self.h, G.ctypes.data, G.shape[1], bs, 0, int(scale), C.byref(out)))
        res["kernel_ms"] = self.lib.rg_s2_last_kernel_ms(self.h)
        return res

    # ---- binary / count traits behind the ABI (rg_s2_bt_*): score test, approximate Firth and saddlepoint corrections ------------------
    def bt_set_null(self, X: np.ndarray, y: np.ndarray, mask: np.ndarray, fitted: np.ndarray, firth_offset=None, passed=None,
                    family: int = 0) -> None:
        """X [C][n], y / mask / fitted / firth_offset [P][n] (sample-fastest): what compute_res_bin leaves per chromosome."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        fitted = np.ascontiguousarray(fitted, dtype=np.float64)
        fo = None if firth_offset is None else np.ascontiguousarray(firth_offset, dtype=np.float64)
        ps = None if passed is None else np.ascontiguousarray(passed, dtype=np.uint8)
        if X.shape != (self.C, self.n) or any(a.shape != (self.P, self.n) for a in (y, mask, fitted)) or (fo is not None and fo.shape != (self.P, self.n)):
            raise ValueError("bt_set_null: expected X %s and [P][n] arrays %s" % ((self.C, self.n), (self.P, self.n)))
        nm = _BtNull(int(family), 0, X.ctypes.data, y.ctypes.data, mask.ctypes.data, fitted.ctypes.data,
                     None if fo is None else fo.ctypes.data, None if ps is None else ps.ctypes.data)
        self._check(self.lib.rg_s2_bt_set_null(self.h, C.byref(nm)))

    def _bt_out(self, bs: int, packed: bool):
        res = {"stats": np.zeros((bs, self.P)), "bhat": np.zeros((bs, self.P)), "denum": np.zeros((bs, self.P)),
               "test_ignored": np.zeros((bs, self.P), np.uint8), "mean": np.zeros(bs), "ignored": np.zeros(bs, np.int32),
               "sparse": np.zeros(bs, np.uint8), "counts": np.zeros((bs, 4), np.int32), "vstat": np.zeros((bs, 4)),
               "total_p": np.zeros((bs, self.P)), "n_obs_p": np.zeros((bs, self.P), np.int32)}
        out = _BtOut(*[res[k].ctypes.data for k in ("stats", "bhat", "denum", "test_ignored", "mean", "ignored", "sparse", "counts", "vstat",
                                                     "total_p", "n_obs_p")])
        return res, out

    def bt_score_packed(self, rows: np.ndarray, flip: bool = False, numtol: float = NUMTOL) -> dict:
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        bs = rows.shape[0]
        res, out = self._bt_out(bs, True)
        self._check(self.lib.rg_s2_bt_score_packed(self.h, rows.ctypes.data, rows.shape[1], bs, 0, 1 if flip else 0, float(numtol), C.byref(out)))
        res["kernel_ms"] = self.lib.rg_s2_last_kernel_ms(self.h)
        return res

    def bt_score_int(self, G: np.ndarray, scale: int, numtol: float = NUMTOL) -> dict:
        G = np.ascontiguousarray(G, dtype=np.uint16)
        bs = G.shape[0]
        res, out = self._bt_out(bs, False)
        self._check(self.lib.rg_s2_bt_score_int(self.h, G.ctypes.data, G.shape[1], bs, 0, int(scale), float(numtol), C.byref(out)))
        res["kernel_ms"] = self.lib.rg_s2_last_kernel_ms(self.h)
        return res

    def bt_correct(self, kind: int, variant, trait, fast, firth_se: bool = False) -> dict:
        """Corrections of the flagged (variant, trait) pairs of the block last scored; kind = BT_FIRTH_APPROX or BT_SPA."""
        variant = np.ascontiguousarray(variant, dtype=np.int32)
        trait = np.ascontiguousarray(trait, dtype=np.int32)
        fast = np.ascontiguousarray(fast, dtype=np.uint8)
        k = variant.size
        arr = (_BtCorr * max(1, k))()
        self._check(self.lib.rg_s2_bt_correct(self.h, int(kind), k, variant.ctypes.data, trait.ctypes.data, fast.ctypes.data, 1 if firth_se else 0, arr))
        return {"beta": np.array([arr[i].beta for i in range(k)]), "se": np.array([arr[i].se for i in range(k)]),
                "chisq": np.array([arr[i].chisq for i in range(k)]), "logp": np.array([arr[i].logp for i in range(k)]),
                "fail": np.array([arr[i].fail for i in range(k)], np.int32), "kernel_ms": self.lib.rg_s2_last_kernel_ms(self.h)}
