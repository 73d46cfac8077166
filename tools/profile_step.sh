# One rocprofv3 kernel trace of bench.py's cfg-2 steps -> timeline + launches under 60 us (a quick look between changes).
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/profile_step.sh tag'
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-p}
mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -o r4 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-layer-gemms --no-also --no-workloads > $O/prof.log 2>&1)
python3 tools/timeline.py $O/prof/r4_results.db > $O/timeline.txt 2>&1; tail -1 $O/timeline.txt
python3 tools/small_launches.py $O/prof/r4_results.db > $O/small_launches.txt; tail -1 $O/small_launches.txt
python3 tools/db_to_stats.py $O/prof/r4_results.db $O/kernel_stats.csv | tail -1
python3 tools/step_kernels.py $O/prof/r4_results.db > $O/step_kernels.txt
rm -f $O/prof/*.db
