"""Every asr_gemm_f32 call of one cfg-2 train step (shape, layout, epilogue, whether the 256 x 128 LDS-DMA kernel takes
it), each timed on its own with cold operands, sorted by time: where the GEMM time of the step goes."""
import os, sys, collections, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd', ROOT + '/tests/golden']
import numpy as np, torch
import __graft_entry__ as entry
entry.build()
import bench, synth, parallel, model as M, hip_backend as hb
from parallel import FlatAdam
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
for _ in range(2): step()
torch.cuda.synchronize()
lib = hb.load()
real = lib.asr_gemm_f32
calls = []
class Spy(object):
    def __call__(self, ta, tb, M_, N, K, A, lda, B, ldb, C, ldc, bias, relu, acc, batch, sA, sB, sC, sk, ar, st):
        v = lambda x: int(getattr(x, 'value', x) or 0)
        calls.append((int(ta), int(tb), int(M_), int(N), int(K), int(batch), v(bias) != 0, int(relu), int(acc), int(sk), int(lda), int(ldb), int(ldc)))
        return real(ta, tb, M_, N, K, A, lda, B, ldb, C, ldc, bias, relu, acc, batch, sA, sB, sC, sk, ar, st)
lib.asr_gemm_f32 = Spy()
step(); torch.cuda.synchronize()
lib.asr_gemm_f32 = real
flush = torch.empty(256 * 1024 * 1024, device=dev)
agg = collections.OrderedDict()
for c in calls: agg[c] = agg.get(c, 0) + 1
rows = []
for (ta, tb, M_, N, K, batch, bias, relu, acc, sk, lda, ldb, ldc), n in agg.items():
    if batch != 1:
        rows.append((0.0, 'batched x%d %s%s M%d N%d K%d  (x%d calls, not timed)' % (batch, 'T' if ta else 'N', 'T' if tb else 'N', M_, N, K, n))); continue
    A = torch.randn((K, lda) if ta else (M_, lda), device=dev); B = torch.randn((N, ldb) if tb else (K, ldb), device=dev)
    out = torch.zeros(M_, ldc, device=dev); bv = torch.randn(N, device=dev) if bias else None
    Av = A[:, :M_] if ta else A[:, :K]; Bv = B[:, :K] if tb else B[:, :N]
    res = []
    base = os.environ.get('GS_ARITH', 'bf16x6')
    for mode in (base, base + '+narrow', base + '+wide', base + '+sp'):
        ts = []
        for _ in range(4):
            flush.fill_(1.0); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); hb.gemm(Av, Bv, trans_a=bool(ta), trans_b=bool(tb), bias=bv, relu=bool(relu), out=out[:, :N], accumulate=bool(acc), split_k=sk, arith=mode); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        res.append(sorted(ts)[1])
    wide = M_ % 256 == 0 and N % 128 == 0 and K % 32 == 0
    rows.append((res[0] * n, '%s%s M%6d N%5d K%6d sk%2d %s%s%s x%d: %6.0f us policy / %6.0f narrow / %6.0f wide / %6.0f sp each  (%5.1f TF)%s' % (
        'T' if ta else 'N', 'T' if tb else 'N', M_, N, K, sk, 'b' if bias else '-', 'r' if relu else '-', 'a' if acc else '-', n, res[0], res[1], res[2], res[3],
        2.0 * M_ * N * K / res[0] / 1e6, '  [conforms]' if wide else '')))
tot = sum(r[0] for r in rows)
for t, s in sorted(rows, key=lambda r: -r[0]): print(s)
print('sum over the step with the wide kernel enabled: %.0f us' % tot)
