# Round-end measurement pass (one gpurun call): tests, bench lines, rocprofv3 kernel stats, PMC passes, tool outputs.
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/run_round_measurements.sh r5f r05'
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5f}; R=${2:-r05}
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 300 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err && tail -1 $O/bench_cfg2.err &&
for T in 200 400 1600; do timeout -k 10 300 python bench.py --frames $T --no-cpu-baseline --no-layer-gemms > $O/bench_cfg2_T$T.json 2> $O/bench_T$T.err && tail -1 $O/bench_T$T.err || exit 1; done &&
timeout -k 10 300 python bench.py --config cfg1 > $O/bench_cfg1.json 2> $O/bench_cfg1.err && tail -1 $O/bench_cfg1.err &&
timeout -k 10 300 python bench.py --config cfg5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err && tail -1 $O/bench_cfg5.err &&
timeout -k 10 300 python bench.py --scaling strong --global-batch 256 --steps 5 --warmup 2 --no-cpu-baseline --no-layer-gemms > $O/bench_strong_n1.json 2> $O/bench_strong.err && tail -1 $O/bench_strong.err &&
(cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -o r4 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-layer-gemms --no-also --no-workloads > $O/prof.log 2>&1; python3 tools/db_to_stats.py $O/prof/r4_results.db $O/kernel_stats.csv) &&
(cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python3 tools/pmc_probe.py > $O/pmc_f.log 2>&1; timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python3 tools/pmc_probe.py > $O/pmc_w.log 2>&1; python3 tools/pmc_summary.py $O/pmc_f/f_counter_collection.csv $O/pmc_w/w_counter_collection.csv $O/${R}_pmc_lstm_persist.json > $O/pmc_summary.log 2>&1; tail -3 $O/pmc_summary.log) &&
(cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && for A in bf16x6 bf16x3 f32; do timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/mfma_$A -o m -- python3 tools/pmc_mfma_probe.py $A > $O/mfma_$A.log 2>&1; python3 tools/pmc_mfma_summary.py $O/mfma_$A/m_counter_collection.csv $A $O/${R}_pmc_mfma.json > $O/mfma_sum_$A.log 2>&1; tail -1 $O/mfma_$A.log; done) &&
(python3 tools/timeline.py $O/prof/r4_results.db > $O/timeline.txt 2>&1; tail -1 $O/timeline.txt; python3 tools/step_kernels.py $O/prof/r4_results.db > $O/step_kernels.txt 2>&1; timeout -k 10 300 python3 tools/gemm_shapes.py > $O/gemm_shapes.txt 2>&1; tail -1 $O/gemm_shapes.txt; timeout -k 10 200 python3 tools/persist_bench.py 400 > $O/persist_bench.txt 2>&1; tail -2 $O/persist_bench.txt; timeout -k 10 300 python3 tools/workload_times.py > $O/workload_times.log 2>&1; tail -8 $O/workload_times.log) &&
(cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_ssl -o ssl -- python3 tools/workload_times.py ssl judge > $O/prof_ssl.log 2>&1; python3 tools/db_to_stats.py $O/prof_ssl/ssl_results.db $O/ssl_judge_kernel_stats.csv | tail -1) &&
(timeout -k 10 300 python3 tools/gemm_sp_bench.py > $O/gemm_sp_bench.txt 2>&1; tail -4 $O/gemm_sp_bench.txt; timeout -k 10 200 python3 tools/gemm_k80_check.py > $O/gemm_k80.txt 2>&1; tail -3 $O/gemm_k80.txt; python3 tools/small_launches.py $O/prof/r4_results.db > $O/small_launches.txt; tail -1 $O/small_launches.txt) &&
(bash tools/profile_epoch.sh $(basename $O) | tail -1; ASR_ENCODER_ROWS=padded timeout -k 10 300 python bench.py --no-cpu-baseline --no-layer-gemms --no-also --no-workloads > $O/bench_cfg2_padded_rows.json 2> $O/bench_padded.err; tail -1 $O/bench_padded.err) &&
(rm -f $O/prof/*.db $O/prof_ssl/*.db; du -sh $O | tail -1)
