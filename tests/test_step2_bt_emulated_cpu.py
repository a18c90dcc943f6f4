"""The device code of regenie_amd/csrc/step2_bt.hip EXECUTED IN THIS CONTAINER: the source is compiled for the host against tests/hipcpu (a stand-in
for the HIP runtime that runs a kernel one workgroup at a time, its work-items as fibers: __syncthreads, 64-lane __shfl_down and __shared__ arrays
behave as on the device), with plain-loop stand-ins for the i8 contraction entries of step2_qt.hip it calls (tests/hipcpu/s2_contract_host.cpp).
What runs is the library's own rg_s2_bt_set_null / rg_s2_bt_score_* / rg_s2_bt_correct with k_bt_prep, k_bt_firth1, k_bt_spa and k_bt_count_two as
written -- so the bodies of the GPU tests of tests/test_step2_bt_gpu.py (score test, approximate Firth, saddlepoint, full and carriers-only forms, and
the reference's minor-allele flip) can be held to the oracle without a GPU.  It does not replace those tests: the compiler, the matrix-core
contraction and the memory system of the device are not under test here, the arithmetic and the control flow of these kernels are."""
import ctypes as C
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emulated(tmp_path_factory):
    d = tmp_path_factory.mktemp("s2emu")
    so = str(d / "libs2emu.so")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-w", "-shared", "-x", "c++", "-I" + os.path.join(ROOT, "tests", "hipcpu"),
                        os.path.join(ROOT, "regenie_amd", "csrc", "step2_bt.hip"), os.path.join(ROOT, "tests", "hipcpu", "s2_contract_host.cpp"), "-o", so],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lib = C.CDLL(so)
    lib.rg_s2_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int32, C.c_int32]
    lib.rg_s2_destroy.argtypes = [C.c_void_p]
    lib.rg_s2_destroy.restype = None
    lib.rg_s2_last_error.argtypes = [C.c_void_p]
    lib.rg_s2_last_error.restype = C.c_char_p
    lib.rg_s2_set_sparse_rule.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_int32]
    lib.rg_s2_bt_set_null.argtypes = [C.c_void_p, C.c_void_p]
    lib.rg_s2_bt_score_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_void_p]
    lib.rg_s2_bt_score_int.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_void_p]
    lib.rg_s2_bt_correct.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.rg_s2_last_kernel_ms.argtypes = [C.c_void_p]
    lib.rg_s2_last_kernel_ms.restype = C.c_double
    return lib


@pytest.fixture()
def gpu_tests(emulated, monkeypatch):
    import regenie_amd.step2 as step2
    monkeypatch.setattr(step2, "load_library", lambda: emulated)
    from tests import test_step2_bt_gpu as t
    return t


@pytest.mark.parametrize("route", ["packed", "int"])
def test_score_and_corrections_on_the_emulated_device(gpu_tests, route):
    gpu_tests.test_bt_score_and_corrections_against_the_oracle(route)          # the GPU test's own body, at its own size (3,001 samples, 96 pairs x 2 forms x 2 corrections)


@pytest.mark.parametrize("route", ["packed", "int"])
def test_minor_allele_flip_on_the_emulated_device(gpu_tests, route):
    gpu_tests.test_bt_corrections_on_the_allele_the_reference_tests(route)


def test_usage_errors_on_the_emulated_device(gpu_tests):
    gpu_tests.test_bt_usage_errors()
