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
