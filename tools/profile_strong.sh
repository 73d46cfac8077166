# rocprofv3 kernel trace of 256 utterances on ONE GPU (the N = 1 base of the strong-scaling curve) -> kernel stats.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-strong}
mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof -o s -- python3 bench.py --steps 4 --warmup 2 --scaling strong --global-batch 256 --no-cpu-baseline --no-layer-gemms --no-also --no-workloads > $O/bench.json 2> $O/prof.log)
python3 tools/db_to_stats.py $O/prof/s_results.db $O/kernel_stats.csv | tail -1
python3 tools/timeline.py $O/prof/s_results.db > $O/timeline.txt 2>&1; tail -1 $O/timeline.txt
rm -f $O/prof/*.db
