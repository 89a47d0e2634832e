import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _torch_before_the_library():
    """torch ships its own HIP runtime; libmuscato_hip.so links the system one.  Both work in one
    process when torch's runtime opens the device first (bench.py, smoke and the driver's test
    order do that); a test selection that reaches the library first and torch later ("-k scale
    or full") makes torch report "no ROCm-capable device".  Fix the order for every selection."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    yield
