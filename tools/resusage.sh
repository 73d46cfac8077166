#!/bin/bash
# Per-kernel register / LDS / scratch usage of one HIP source (compiler remarks), one line per kernel:
#   tools/resusage.sh semi-supervised-asr_amd/csrc/gemm.hip [extra hipcc flags]
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$src" -o /dev/null -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
  python3 -c '
import re, subprocess, sys
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*) \[-Rpass", line)
    if not m:
        if "error" in line or "warning:" in line: sys.stderr.write(line)
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = [t.split(":", 1)[1].strip()]
        rows.append(cur)
    elif cur and any(t.startswith(k) for k in ("VGPRs:", "AGPRs:", "VGPRs Spill", "ScratchSize", "Occupancy", "LDS Size")):
        cur.append(t.replace(" [bytes/lane]", "").replace(" [bytes/block]", "").replace(" [waves/SIMD]", ""))
names = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.split("\n")
for r, n in zip(rows, names):
    n = n.replace("(anonymous namespace)::", "").split("(")[0]
    print("%-60s %s" % (n[:60], " | ".join(r[1:])))
'
