import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "semi-supervised-asr_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _cpu_share():
    """CPU threads for the oracle / reference-side math of the tests: the affinity mask, capped by the cgroup quota and by 16.
    A GPU box may report the whole host (256 cores, shared with other tenants): torch's default of one thread per core then
    oversubscribes the one GPU's share of cores and the CPU restatements run ten times slower than on 16 threads (bench.py's
    usable_cpus() is the same rule)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    try:
        import torch
        torch.set_num_threads(_cpu_share())
    except Exception:
        pass


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
