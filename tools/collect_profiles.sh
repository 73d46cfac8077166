#!/bin/bash
# Copy the judged summaries of a measurement pass (tools/run_round_measurements.sh <dir>) into profiles/ under the names
# DESIGN.md cites:   bash tools/collect_profiles.sh r4f r04
O=gpurun_out/${1:-r4f}; R=${2:-r04}
for f in cfg2 cfg1 cfg5 cfg2_T200 cfg2_T400 cfg2_T1600 strong_n1 cfg2_padded_rows; do cp $O/bench_$f.json profiles/${R}_bench_$f.json; done
cp $O/kernel_stats.csv profiles/${R}_bench_kernel_stats.csv
cp $O/timeline.txt profiles/${R}_step_timeline.txt
cp $O/epoch_timeline.txt profiles/${R}_epoch_timeline.txt; cp $O/step_kernels.txt profiles/${R}_step_kernels.txt 2>/dev/null
grep -v "^[WE]20[0-9]* \|amdgpu.ids" $O/gemm_shapes.txt > profiles/${R}_gemm_shapes.txt
grep -v "^[WE]20[0-9]* \|amdgpu.ids" $O/persist_bench.txt > profiles/${R}_persist_bench.txt
grep -v "^[WE]20[0-9]* \|amdgpu.ids" $O/workload_times.log > profiles/${R}_workload_times.log
tail -3 $O/tests.log > profiles/${R}_gpu_tests.txt
for f in gemm_sp_bench gemm_k80 small_launches; do grep -v "^[WE]20[0-9]* \|amdgpu.ids" $O/$f.txt > profiles/${R}_$f.txt; done
[ -f $O/ssl_judge_kernel_stats.csv ] && cp $O/ssl_judge_kernel_stats.csv profiles/${R}_ssl_judge_kernel_stats.csv
cp $O/${R}_pmc_mfma.json profiles/${R}_pmc_mfma.json
cp $O/${R}_pmc_lstm_persist.json profiles/${R}_pmc_lstm_persist.json
cp $O/${R}_pmc_FETCH_SIZE_lstm_persist.csv profiles/${R}_pmc_FETCH_SIZE_lstm_persist.csv 2>/dev/null
cp $O/${R}_pmc_WRITE_SIZE_lstm_persist.csv profiles/${R}_pmc_WRITE_SIZE_lstm_persist.csv 2>/dev/null
python3 - $O $R <<'PY'
import csv, sys
O, R = sys.argv[1], sys.argv[2]
# MFMA counter rows of the default arithmetic, one line per kernel (summed over the dispatches of the probe)
rows = list(csv.DictReader(open(O + '/mfma_bf16x6/m_counter_collection.csv')))
agg = {}
for r in rows:
    k = (r['Kernel_Name'][:110], r['Counter_Name'])
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r['Counter_Value'])
with open('profiles/%s_pmc_mfma_bf16x6_counters.csv' % R, 'w') as f:
    f.write('kernel,counter,dispatches,sum\n')
    for (k, c), (n, v) in sorted(agg.items()):
        f.write('"%s",%s,%d,%.0f\n' % (k, c, n, v))
PY
for f in cfg5_side_ab cfg5_timeline solver_run gpu_only_time; do [ -f $O/$f.txt ] && cp $O/$f.txt profiles/${R}_$f.txt; done
ls -la profiles | grep $R
