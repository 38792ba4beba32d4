import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_np():
    import lbm_numpy
    return lbm_numpy


@pytest.fixture(scope="session")
def oracle_c():
    import lbm_c
    lbm_c.build()
    return lbm_c


@pytest.fixture(scope="session")
def pkg():
    import airfoil_cfd_tool_amd
    return airfoil_cfd_tool_amd


def bits_equal(a: np.ndarray, b: np.ndarray) -> bool:
    """Bit-for-bit equality of two float arrays (NaN payloads included)."""
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    u = np.uint32 if a.dtype == np.float32 else np.uint64
    return bool(np.array_equal(a.view(u), b.view(u)))
