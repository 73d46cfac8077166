set -e
cd /root/repo && mkdir -p gpurun_out
true
true
ASR_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --no-workloads > gpurun_out/bench_2p.json 2> gpurun_out/bench_2p.err || { tail -30 gpurun_out/bench_2p.err; exit 1; }
ASR_DIST_BACKEND=gloo ASR_BENCH_INJECT_ABORT=timed timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 1 --no-workloads > gpurun_out/bench_2pi.json 2> gpurun_out/bench_2pi.err || { tail -30 gpurun_out/bench_2pi.err; exit 1; }
grep -h "abort\|repeat" gpurun_out/bench_2pi.err | head
python - <<'PY'
import json
for f in ("gpurun_out/bench_2p.json","gpurun_out/bench_2pi.json"):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["loss"], d["config"]["per_rank"], d["config"]["retimed_after_abort"])
PY
