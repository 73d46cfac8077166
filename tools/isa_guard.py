"""The register-allocator guard: the persistent LSTM / decoder kernels are dependent chains whose time per step
can move by 28 % with identical arithmetic - a prefetch whose value the allocator could not keep in place and copied out of a
temporary right behind its load, or a spill into scratch on the chain.  No numerics test sees that.  This module compiles
csrc/lstm_persist.hip and csrc/dec_persist.hip to gfx950 assembly (hipcc -S: works without a GPU) and reads, per kernel:

  instructions        machine instructions of the function
  scratch_bytes       .amdhsa_private_segment_fixed_size (spills; 0 for every LSTM kernel)
  vgprs / agprs       .amdhsa_next_free_vgpr (unified file on gfx950) and the accumulation offset
  lds_bytes           .amdhsa_group_segment_fixed_size
  tight_x4_loads      16-byte vector loads (global_/buffer_load_dwordx4) that an s_waitcnt vmcnt waits for within
                      TIGHT instructions of their issue, in program order - the signature of 4.5.4's regression

    python tools/isa_guard.py --update      # rewrite tests/golden/isa_table.json from the current sources (a conscious act)
    python tools/isa_guard.py               # compare, print what moved

tests/test_isa_guard_cpu.py asserts the comparison (`-m "not gpu"`, ~1 min of hipcc)."""
import json
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "semi-supervised-asr_amd", "csrc")
TABLE = os.path.join(ROOT, "tests", "golden", "isa_table.json")
SOURCES = ("lstm_persist.hip", "dec_persist.hip", "gemm.hip")
TIGHT = 8                                        # instructions
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only"]     # the flags of build() + -S


def asm_dir():
    d = os.path.join(ROOT, "semi-supervised-asr_amd", "lib", "isa")
    os.makedirs(d, exist_ok=True)
    return d


def compile_asm(src):
    """csrc/<src> -> lib/isa/<src>.s (git-ignored like the library), rebuilt when a source or header is newer."""
    out = os.path.join(asm_dir(), src[:-4] + ".s")
    deps = [os.path.join(CSRC, src), os.path.join(ROOT, "include", "asr_hip.h")] + \
        [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        tmp = "%s.%d.tmp" % (out, os.getpid())
        subprocess.check_call([hipcc if os.path.exists(hipcc) else "hipcc"] + FLAGS + ["-o", tmp, os.path.join(CSRC, src)],
                              stderr=subprocess.DEVNULL)
        os.replace(tmp, out)
    return out


def short_name(mangled):
    m = re.match(r"_ZN12_GLOBAL__N_1\d+(.*?)(?:EEvNS_.*|Ev.*|E[a-z].*)?$", mangled)
    name = m.group(1) if m else mangled
    name = re.sub(r"^(\w+?)I(L.*)$", r"\1<\2", name)
    args = re.findall(r"L([ib])(\d+)E", name)
    if "<" in name:
        name = name.split("<")[0] + "<" + ",".join(("true" if v == "1" else "false") if t == "b" else v for t, v in args) + ">"
    return name


_MEM = re.compile(r"^(global_load|global_store|global_atomic|buffer_load|buffer_store|buffer_atomic|scratch_load|scratch_store|flat_)")
_X4 = re.compile(r"^(global_load_dwordx4|buffer_load_dwordx4)")


def parse(path):
    """-> {kernel: metrics} for every kernel (.amdhsa_kernel) of an assembly file."""
    txt = open(path).read()
    out = {}
    for m in re.finditer(r"^(_Z\w+):\s*(?:;.*)?$", txt, re.M):
        name = m.group(1)
        desc = txt.find(".amdhsa_kernel " + name + "\n", m.end())
        end = txt.find(".Lfunc_end", m.end())
        if end < 0 or desc < 0:
            continue
        end = min(end, desc)                      # (the descriptor sits between the last instruction and .Lfunc_end)
        body = [l.split(";")[0].strip() for l in txt[m.end():end].splitlines()]
        ins = [l for l in body if l and not l.startswith(".") and not l.endswith(":")]
        dtxt = txt[desc:txt.find(".end_amdhsa_kernel", desc)]

        def field(key, default=0):
            mm = re.search(r"\.amdhsa_%s\s+(\d+)" % key, dtxt)
            return int(mm.group(1)) if mm else default
        # vector-memory ops in program order; a wait vmcnt(k) at a point where n have been issued completes ops 0 .. n-k-1
        issued, tight = [], 0                     # (instruction index, is a 16-byte load, already waited for)
        done = 0
        for i, l in enumerate(ins):
            if _MEM.match(l) and not l.startswith(("global_store", "buffer_store", "scratch_store")):
                issued.append((i, bool(_X4.match(l))))
            elif l.startswith("s_waitcnt"):
                mm = re.search(r"vmcnt\((\d+)\)", l)
                if mm:
                    upto = len(issued) - int(mm.group(1))
                    for j in range(done, max(done, upto)):
                        if issued[j][1] and i - issued[j][0] <= TIGHT:
                            tight += 1
                    done = max(done, upto)
        out[short_name(name)] = dict(instructions=len(ins), scratch_bytes=field("private_segment_fixed_size"),
                                     vgprs=field("next_free_vgpr"), agpr_offset=field("accum_offset"),
                                     lds_bytes=field("group_segment_fixed_size"), tight_x4_loads=tight,
                                     mfma=sum(1 for l in ins if l.startswith("v_mfma")))
    return out


def measure():
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as pool:
        paths = list(pool.map(compile_asm, SOURCES))
    table = {}
    for src, path in zip(SOURCES, paths):
        table[src] = parse(path)
    return table


def compare(table, want, instr_tol=0.02):
    """-> list of findings (empty: nothing moved).  Spills and tight loads may not grow, the instruction count may move by 2 %,
    registers by 8; a kernel that is new or gone is a finding (regenerate the table when that is meant)."""
    bad = []
    for src in want:
        got_k, want_k = table.get(src, {}), want[src]
        for k in sorted(set(got_k) | set(want_k)):
            if k not in got_k or k not in want_k:
                bad.append("%s: %s is %s" % (src, k, "new (not in the table)" if k in got_k else "gone"))
                continue
            g, w = got_k[k], want_k[k]
            if g["scratch_bytes"] > w["scratch_bytes"]:
                bad.append("%s: %s spills %d bytes of scratch (table: %d)" % (src, k, g["scratch_bytes"], w["scratch_bytes"]))
            if g["tight_x4_loads"] > w["tight_x4_loads"]:
                bad.append("%s: %s waits for %d 16-byte loads within %d instructions of their issue (table: %d) - a prefetch "
                           "copied out of a temporary? (DESIGN 4.5.4)" % (src, k, g["tight_x4_loads"], TIGHT, w["tight_x4_loads"]))
            if abs(g["instructions"] - w["instructions"]) > instr_tol * w["instructions"]:
                bad.append("%s: %s has %d instructions (table: %d)" % (src, k, g["instructions"], w["instructions"]))
            if g["vgprs"] > w["vgprs"] + 8:
                bad.append("%s: %s uses %d VGPRs (table: %d)" % (src, k, g["vgprs"], w["vgprs"]))
            if g["mfma"] != w["mfma"]:
                bad.append("%s: %s has %d MFMA instructions (table: %d)" % (src, k, g["mfma"], w["mfma"]))
    return bad


def range_checked_loads(kernel_filter=("gemm_bfk_kernel", "gemm_bfs_kernel")):
    """ADVICE r4, statically: the hardware's buffer range check covers the VGPR offset (+ the immediate), not the SGPR offset.
    gemm_bfk_kernel (its M tiles) and the masked-K-tail instantiations of gemm_bfs_kernel (KT = true) reach behind their
    operand by design and rely on that check, so every buffer load of theirs must carry a ZERO SGPR offset.
    -> {kernel: [offending instructions]} over csrc/gemm.hip."""
    path = compile_asm("gemm.hip")
    txt = open(path).read()
    out = {}
    for m in re.finditer(r"^(_Z\w+):\s*(?:;.*)?$", txt, re.M):
        name = m.group(1)
        short = short_name(name)
        if not short.startswith(kernel_filter):
            continue
        if short.startswith("gemm_bfs_kernel") and not short.endswith(",true>"):
            continue                                   # KT = false: every K tile lies inside the operand
        desc = txt.find(".amdhsa_kernel " + name + "\n", m.end())
        if desc < 0:
            continue
        bad = []
        for l in txt[m.end():desc].splitlines():
            l = l.split(";")[0].strip()
            if l.startswith("buffer_load"):
                # buffer_load_dwordx4 vdst, voffset, srsrc[4], soffset [offen] [offset:imm]
                ops = [o.strip() for o in l.split(None, 1)[1].split(",")]
                soff = ops[3].split()[0]
                if soff not in ("0", "off", "null"):
                    bad.append(l)
        out[short] = bad
    return out


if __name__ == "__main__":
    t = measure()
    if "--update" in sys.argv:
        with open(TABLE, "w") as f:
            json.dump(t, f, indent=0, sort_keys=True)
        print("wrote %s: %s" % (TABLE, ", ".join("%s %d kernels" % (s, len(k)) for s, k in t.items())))
    else:
        with open(TABLE) as f:
            findings = compare(t, json.load(f))
        print("\n".join(findings) if findings else "nothing moved")
        sys.exit(1 if findings else 0)
