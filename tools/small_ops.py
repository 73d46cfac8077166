"""Which Python lines launch the small kernels of one cfg-2 train step: torch profiler with stacks, aten ops that launch a
device kernel grouped by the innermost frame inside this repo."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd', ROOT + '/tests/golden']
import numpy as np, torch
import __graft_entry__ as entry
entry.build()
import bench, synth, parallel, model as M, hip_backend as hb
from parallel import FlatAdam
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda')
spec = bench.CONFIGS[os.environ.get('CFG', 'cfg2')]
cfg = dict(spec['model'])
net = M.E2E(labeldist=synth.labeldist(cfg['output_dim'], 5), **cfg)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.e2e_weights(cfg, 99).items()})
net = net.to(dev).train()
opt = FlatAdam(net, lr=5e-4, weight_decay=1e-6, amsgrad=True, max_grad_norm=5.0)
xs, lens, ys = synth.ragged_batch(spec['batch'], spec['frames'], cfg['input_dim'], cfg['output_dim'], 1234)
xs_r, lens_r, ys_r, info = parallel.shard_batch(xs, lens, ys, 0, 1)
xs_d = torch.from_numpy(np.ascontiguousarray(xs_r)).to(dev); ys_d = [torch.from_numpy(y).to(dev) for y in ys_r]
tl = M.padded_lengths(info['t_max'], cfg['enc_n_layers'], cfg['subsample'])
def step():
    _, lp, _, _ = net(xs_d, lens_r, ys_d, tf_rate=1.0, total_length=tl, olength=info['olength'])
    loss = parallel.local_loss(lp, info); opt.zero_grad(); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
evs = [ev for ev in prof.events() if ev.device_type == torch.autograd.DeviceType.CPU and ev.kernels and not any(c.kernels for c in ev.cpu_children)]
evs.sort(key=lambda e: e.time_range.start)
tot = 0; n = 0
for ev in evs:
    d = sum(k.duration for k in ev.kernels)
    par = ev.cpu_parent.name if ev.cpu_parent is not None else '-'
    big = d > 60
    if not big: tot += d; n += len(ev.kernels)
    print('%8.1f us %2d k  %-26s in %-34s %s' % (d, len(ev.kernels), ev.name[:26], par[:34], str(ev.input_shapes)[:90]))
print('small ops (<= 60 us): %.0f us in %d kernels' % (tot, n))
