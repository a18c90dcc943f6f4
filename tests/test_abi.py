"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, and exports every symbol
include/*.h declares; without a GPU the product path fails loudly instead of falling back."""
import ctypes
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from regenie_amd import build
    path = build.build()
    return ctypes.CDLL(path)


def _declared():
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names += re.findall(r"\b(rg_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_every_declared_symbol_is_exported(lib):
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "missing export " + n


def test_engine_exports_match_header():
    from regenie_amd import engine
    assert sorted(engine.EXPORTS) == _declared()


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from regenie_amd.engine import RgError, Step1Engine
    with pytest.raises(RgError):
        Step1Engine(0)


def test_product_does_not_import_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "regenie_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                s = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle", s, flags=re.M), f
