"""Host time of one train step: how long Solver.sup_train_one_iteration takes to ENQUEUE a cfg-2 step when the GPU queue is
empty (a synchronise in front of every call), next to the step's GPU time.  The step is GPU-bound only while the first stays
below the second."""
import os, sys, time, tempfile, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd', ROOT + '/tests/golden']
import numpy as np, torch
import __graft_entry__ as entry
entry.build()
import bench, synth
dev = torch.device('cuda', 0)
spec = bench.CONFIGS[os.environ.get('CFG', 'cfg2')]
cfg, B, T = dict(spec['model']), spec['batch'], spec['frames']
with contextlib.redirect_stdout(sys.stderr):
    sv = bench.make_solver(cfg, B, T, tempfile.mkdtemp())
xs, lens, ys = synth.ragged_batch(B, T, cfg['input_dim'], cfg['output_dim'], 1234)
xs_d, ys_d = torch.from_numpy(xs).to(dev), [torch.from_numpy(y).to(dev) for y in ys]
for _ in range(3):
    sv.sup_train_one_iteration(xs_d, lens, ys_d, 1.0)
sv.flush(); torch.cuda.synchronize()
host, gpu = [], []
for _ in range(15):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    sv.sup_train_one_iteration(xs_d, lens, ys_d, 1.0)
    e1.record(); host.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize(); gpu.append(e0.elapsed_time(e1))
sv.flush()
print("host enqueue %.2f ms (min %.2f) | first kernel to last kernel %.2f ms (min %.2f)" % (np.median(host), min(host), np.median(gpu), min(gpu)))
if os.environ.get("PROFILE"):
    import cProfile, pstats
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(int(os.environ.get("PROFILE_STEPS", "5"))):
        sv.sup_train_one_iteration(xs_d, lens, ys_d, 1.0)
    pr.disable(); sv.flush()
    pstats.Stats(pr).sort_stats(os.environ.get("SORT", "cumulative")).print_stats(int(os.environ.get("TOP", "45")))
