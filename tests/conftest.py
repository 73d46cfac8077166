import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "semi-supervised-asr_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _fresh_persistent_probation():
    """A test that sent the process to the per-step kernels (hip_backend.disable_persistent) must not leave a probation
    behind that ends in the middle of a later test."""
    yield
    hb = sys.modules.get("hip_backend")
    if hb is not None and hasattr(hb, "_PROBATION"):
        hb._PROBATION.update(wanted=None, aborts=0, steps=0, retry_at=None)
