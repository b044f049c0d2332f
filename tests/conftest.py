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


@pytest.fixture(autouse=True)
def _extra_library_options(monkeypatch):
    """BHS_TEST_OPTS=key=value,..: extra library options for every multiply of the parity tests (e.g.
    class_numeric=1 runs the whole suite on the workgroup form of the class kernel)."""
    extra = os.environ.get("BHS_TEST_OPTS", "")
    if not extra:
        yield
        return
    from benchmark_spgemm_using_csr_amd import facade
    add = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in extra.split(",") if kv}
    real = facade.bhsparse.initPlatform

    def init(self, *a, **kw):
        err = real(self, *a, **kw)
        if err == 0:
            for k_, v_ in add.items():
                self.set_option(k_, v_)
        return err
    monkeypatch.setattr(facade.bhsparse, "initPlatform", init)
    yield
