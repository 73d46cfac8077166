# Same-box A/B of the cfg-2 bench: the tree in tmp_pre/ (an older commit, built there) against this tree, interleaved.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/ab_bench.sh tag [rounds] [ENV=VALUE ...for the post runs]'
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-ab}; N=${2:-3}; shift; shift
mkdir -p $O
one() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3))"; }
A="--steps 30 --warmup 5 --no-cpu-baseline --no-layer-gemms --no-also --no-workloads"
for i in $(seq $N); do
  (cd tmp_pre && timeout -k 10 200 python bench.py $A 2>/dev/null | one pre) >> $O/ab.txt
  (timeout -k 10 200 python bench.py $A 2>/dev/null | one post) >> $O/ab.txt
  for kv in "$@"; do (env $kv timeout -k 10 200 python bench.py $A 2>/dev/null | one "post[$kv]") >> $O/ab.txt; done
done
cat $O/ab.txt
