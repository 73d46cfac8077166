# rocprofv3 kernel trace of the product's epoch loop (bench.py --epoch-only: Solver.sup_train_one_epoch through the input
# pipeline) -> two consecutive steps' timeline with the idle gaps (profiles/r05_epoch_timeline.txt).
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/profile_epoch.sh tag'
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-ep}
mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -o ep -- python3 bench.py --epoch-only 30 > $O/epoch_profiled.json 2> $O/prof.log)
python3 tools/timeline.py $O/prof/ep_results.db 2 > $O/epoch_timeline.txt 2>&1; tail -1 $O/epoch_timeline.txt
rm -f $O/prof/*.db
