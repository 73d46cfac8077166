"""The register-allocator guard as a test that needs no GPU (tools/isa_guard.py): csrc/lstm_persist.hip and
csrc/dec_persist.hip are compiled to gfx950 assembly with the flags of build() and every persistent kernel's instruction count,
scratch (spill) bytes, register count, MFMA count and the number of 16-byte loads that are waited for right behind their issue
are held to tests/golden/isa_table.json.  A change that is meant moves the table with it:
`python tools/isa_guard.py --update` - and says so in its commit."""
import json
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def tables():
    import isa_guard
    if not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")):
        pytest.skip("no hipcc")
    with open(isa_guard.TABLE) as f:
        want = json.load(f)
    return isa_guard, isa_guard.measure(), want


def test_persistent_kernels_compile_to_the_committed_shape(tables):
    guard, got, want = tables
    findings = guard.compare(got, want)
    assert not findings, "\n".join(findings)


def test_the_chains_do_not_spill(tables):
    """Every LSTM persistent kernel of the encoder's widths: no scratch at all (the H = 640 forward of the judge spills 5 - 7
    dwords: held to the table).  The decoder's forward kernels: none; its backward kernels are the ones that spill (57 dwords in
    the 4-slice D = 512 instantiation) - held to the table until that is fixed."""
    guard, got, want = tables
    for name, m in got["lstm_persist.hip"].items():
        if "<640," not in name:
            assert m["scratch_bytes"] == 0, name
    for name, m in got["dec_persist.hip"].items():
        if name.startswith("dec_persist_fwd_kernel") or name.startswith("att_m"):
            assert m["scratch_bytes"] == 0, name


def test_the_guard_sees_a_planted_regression(tables):
    """The comparison itself: a kernel whose table entry is tightened by one tight load / 4 bytes of scratch / 3 % of its
    instructions is reported."""
    guard, got, want = tables
    k = "lstm_persist_bwd_rs_kernel<512,8,3,true>"
    assert k in got["lstm_persist.hip"], sorted(got["lstm_persist.hip"])[:5]
    for field, delta in (("tight_x4_loads", -1), ("scratch_bytes", -4), ("instructions", -int(0.03 * got["lstm_persist.hip"][k]["instructions"]))):
        fake = json.loads(json.dumps(want))
        fake["lstm_persist.hip"][k][field] = got["lstm_persist.hip"][k][field] + delta
        assert any(k in f for f in guard.compare(got, fake)), field


def test_kernels_that_rely_on_the_buffer_range_check_load_with_a_zero_sgpr_offset():
    """ADVICE r4 (the static half; the dynamic half is test_gemm_operands_flush_against_the_end_of_an_allocation): the range
    check of a buffer instruction sees the VGPR offset, not the SGPR offset.  gemm_bfk_kernel and the KT instantiations of
    gemm_bfs_kernel fetch behind their operand by design - every buffer load of theirs carries its whole offset in the VGPR."""
    import isa_guard
    if not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")):
        pytest.skip("no hipcc")
    found = isa_guard.range_checked_loads()
    assert any(k.startswith("gemm_bfk_kernel") for k in found) and any(k.startswith("gemm_bfs_kernel") for k in found), sorted(found)
    offending = {k: v[:3] for k, v in found.items() if v}
    assert not offending, offending
