"""Which Python lines launch the small kernels of one cfg-2 train step (Solver.sup_train_one_iteration, the call bench.py
times): torch profiler with stacks, every device kernel of one step in launch order with the aten op / autograd node that
launched it and the innermost frame inside this repo."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd', ROOT + '/tests/golden']
import numpy as np, torch
import __graft_entry__ as entry
entry.build()
import bench, synth
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda')
name = os.environ.get('CFG', 'cfg2')
spec = bench.CONFIGS[name]
cfg = dict(spec['model'])
solver = bench.make_solver(cfg, spec['batch'], spec['frames'], tempfile.mkdtemp())
xs, lens, ys = synth.ragged_batch(spec['batch'], spec['frames'], cfg['input_dim'], cfg['output_dim'], 1234)
xs_d = torch.from_numpy(np.ascontiguousarray(xs)).to(dev)
ys_d = [torch.from_numpy(y).to(dev) for y in ys]
lens = [int(v) for v in lens]


def step():
    return solver.sup_train_one_iteration(xs_d, lens, ys_d, 1.0)


for _ in range(4):
    step()
solver.flush()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
solver.flush()


def frame_of(ev):
    for fr in (ev.stack or []):
        if ROOT in fr and '/tools/' not in fr:
            return fr.replace(ROOT + '/', '').replace('semi-supervised-asr_amd/', '')
    return '-'


evs = [ev for ev in prof.events() if ev.device_type == torch.autograd.DeviceType.CPU and ev.kernels
       and not any(c.kernels for c in ev.cpu_children)]
evs.sort(key=lambda e: e.time_range.start)
tot = 0.0
n = 0
for ev in evs:
    d = sum(k.duration for k in ev.kernels)
    par = ev
    while par.cpu_parent is not None and not par.name.endswith('Backward') and 'autograd::' not in par.name:
        par = par.cpu_parent
    if d <= 60:
        tot += d
        n += len(ev.kernels)
    print('%8.1f us %2d k  %-30s %-30s %s' % (d, len(ev.kernels), ev.name[:30], par.name[:30] if par is not ev else '', frame_of(ev)[:70]))
print('small ops (<= 60 us): %.0f us in %d kernels' % (tot, n))
