"""Per-step time of the cfg-2 Solver step from one device event per step (no host sync inside the loop): median, max and the
outliers - what a host-side pause (garbage collection, the autograd thread) costs when it happens.  argv[1] = steps,
JIT_GC=freeze|disable|default."""
import os, sys, gc, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd', ROOT + '/tests/golden']
import numpy as np, torch
import __graft_entry__ as entry
entry.build()
import bench, synth
dev = torch.device('cuda')
spec = bench.CONFIGS['cfg2']
cfg = dict(spec['model'])
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
tmp = tempfile.mkdtemp(prefix='jit_')
import contextlib
with contextlib.redirect_stdout(sys.stderr):
    sv = bench.make_solver(cfg, spec['batch'], spec['frames'], tmp)
xs, lens, ys = synth.ragged_batch(spec['batch'], spec['frames'], cfg['input_dim'], cfg['output_dim'], 1234)
xs_d = torch.from_numpy(np.ascontiguousarray(xs)).to(dev); ys_d = [torch.from_numpy(y).to(dev) for y in ys]
mode = os.environ.get('JIT_GC', 'default')
with contextlib.redirect_stdout(sys.stderr):
    for _ in range(5): sv.sup_train_one_iteration(xs_d, lens, ys_d, 1.0)
    sv.flush(); torch.cuda.synchronize()
    if mode == 'freeze': gc.collect(); gc.freeze()
    if mode == 'disable': gc.disable()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for i in range(steps):
        sv.sup_train_one_iteration(xs_d, lens, ys_d, 1.0)
        ev[i + 1].record()
    sv.flush(); torch.cuda.synchronize()
d = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(steps)])
print('gc %-8s steps %d: mean %.3f ms, median %.3f, p95 %.3f, max %.3f; steps over median + 0.3 ms: %d (%s)' % (
    mode, steps, d.mean(), np.median(d), np.percentile(d, 95), d.max(), int((d > np.median(d) + 0.3).sum()),
    ' '.join('%.2f' % v for v in sorted(d)[-5:])), flush=True)
