#!/bin/bash
# GPU-only time of a step (sum of kernel durations per step) next to its wall time: how host-bound is a workload?
#   bash tools/gpu_only_time.sh <outdir> "--config cfg1" | "--frames 200"
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -o g -- python3 bench.py $@ --steps 20 --warmup 5 --no-cpu-baseline --no-layer-gemms --no-also --no-workloads > $O/bench.json 2> $O/prof.log)
python3 tools/db_to_stats.py $O/prof/g_results.db $O/kernel_stats.csv > /dev/null
timeout -k 10 200 python3 bench.py $@ --steps 20 --warmup 5 --no-cpu-baseline --no-layer-gemms --no-also --no-workloads > $O/bench_plain.json 2> /dev/null
python3 - $O <<'PY'
import csv, json, sys
O = sys.argv[1]
rows = list(csv.DictReader(open(O + "/kernel_stats.csv")))
tot = sum(float(r["TotalDurationUs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
d = json.loads(open(O + "/bench.json").read().strip().splitlines()[-1])
p = json.loads(open(O + "/bench_plain.json").read().strip().splitlines()[-1])
print("wall %.3f ms/step (%.3f under the profiler) | sum of kernel durations %.3f ms/step, %d launches/step" % (p["ms_per_step"], d["ms_per_step"], tot / 25 / 1e3, calls / 25))
PY
rm -f $O/prof/*.db
