import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
EXAMPLE = os.path.join(GOLDEN, "example")


# the C++ driver frees its contexts and tears the runtime down in order under the tests (its default is to leave through _exit once
# every output file is closed): leaks and use-after-free surface here instead of being hidden by the fast exit
os.environ.setdefault("RG_TEARDOWN", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def example_dir():
    return EXAMPLE


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_first():
    """On the GPU box torch must bring up its HIP runtime before librg_step1_hip.so does: initialised the other way round,
    torch.cuda reports "No HIP GPUs are available" for the rest of the process (bench.py imports torch first as well)."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    yield
