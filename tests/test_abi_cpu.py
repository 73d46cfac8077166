"""CPU-side checks: the C-ABI library builds/loads and exports every symbol include/asr_hip.h declares;
the product refuses to compute without a GPU (no CPU fallback)."""
import os
import re

import pytest
import torch

import __graft_entry__ as entry

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_build_and_exports():
    entry.build()
    import hip_backend as hb
    lib = hb.load()
    header = open(os.path.join(ROOT, "include", "asr_hip.h")).read()
    declared = set(re.findall(r"^(?:int|void\*?)\s+(asr_\w+)\s*\(", header, flags=re.M))
    assert declared, "no declarations parsed"
    assert declared == set(hb.EXPORTS), (declared ^ set(hb.EXPORTS))
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.asr_abi_version() == hb.ABI_VERSION


def test_no_cpu_fallback():
    entry.build()
    import hip_backend as hb
    a = torch.zeros(4, 4)
    with pytest.raises(RuntimeError):
        hb.gemm(a, a)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "semi-supervised-asr_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("# oracle", ""), fn


def test_persistent_kernels_come_back_after_a_probation(monkeypatch):
    """Host logic of hip_backend.disable_persistent / persistent_step_tick: an abort sends the process to the per-step
    kernels for PERSIST_RETRY_STEPS train steps, twice as long after every further abort; a permanent switch (several
    processes on one card) stays; 0 steps = never again."""
    import hip_backend as hb
    monkeypatch.setattr(hb, "_PROBATION", dict(wanted=None, aborts=0, steps=0, retry_at=None))
    monkeypatch.setattr(hb, "PERSIST_RETRY_STEPS", 3)
    for name in ("USE_PERSIST", "USE_PERSIST_DEC", "USE_PERSIST_DEC_BWD"):
        monkeypatch.setattr(hb, name, True)
    monkeypatch.setattr(hb, "USE_PERSIST_DEC_BWD", False)            # a flag that was off at the start stays off
    assert not hb.persistent_step_tick() and hb.persistent_probation() == (0, None)
    hb.disable_persistent()
    assert not (hb.USE_PERSIST or hb.USE_PERSIST_DEC or hb.USE_PERSIST_DEC_BWD) and hb.persistent_probation() == (1, 3)
    assert [hb.persistent_step_tick() for _ in range(3)] == [False, False, True]
    assert (hb.USE_PERSIST, hb.USE_PERSIST_DEC, hb.USE_PERSIST_DEC_BWD) == (True, True, False)
    assert hb.persistent_probation() == (1, None)
    hb.disable_persistent()                                          # the second abort: twice as long
    assert hb.persistent_probation() == (2, 6)
    assert [hb.persistent_step_tick() for _ in range(6)] == [False] * 5 + [True] and hb.USE_PERSIST
    monkeypatch.setattr(hb, "PERSIST_RETRY_STEPS", 0)
    hb.disable_persistent()
    assert hb.persistent_probation() == (3, None) and not any(hb.persistent_step_tick() for _ in range(50))
    assert not hb.USE_PERSIST
    monkeypatch.setattr(hb, "PERSIST_RETRY_STEPS", 3)
    hb.disable_persistent(permanent=True)                            # a shared card: for good
    assert not any(hb.persistent_step_tick() for _ in range(50)) and not hb.USE_PERSIST
    hb.disable_persistent()                                          # ... whatever happens later
    assert hb.persistent_probation()[1] is None
