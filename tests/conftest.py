import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def hiplib():
    """Builds (if stale) and loads libbhsparse_hip.so. No fallback."""
    from benchmark_spgemm_using_csr_amd import _lib
    if not (os.path.exists(_lib.SO_PATH) and os.path.exists(_lib.SO_PATH_F32)):
        _lib.build()
    return _lib.load()


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: z[k] for k in z.files}
