import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def golden():
    return load_golden


# ---- guard-band mode (RLPYT_CANARY=1): every CUDA buffer this package allocates is surrounded by
# 0xFF guards (rlpyt_amd/utils/canary.py) and the guards are verified after every test ------------
_CANARY = os.environ.get("RLPYT_CANARY", "0") == "1"
if _CANARY:
    from rlpyt_amd.utils import canary as _canary
    _canary.enable()


@pytest.fixture(autouse=True)
def _canary_guard(request):
    yield
    if _CANARY and request.node.get_closest_marker("gpu") is not None:
        _canary.check(f"after {request.node.nodeid}")
