"""Times the persistent LSTM sequence kernels (cfg-2 layer shape B=32, H=512, both directions) for the shipped
library and any measurement variants under scratchlibs/."""
import ctypes, sys, os, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd']
import torch, numpy as np
import hip_backend as hb
dev = torch.device('cuda')
H, B, T = int(os.environ.get("PB_H", "512")), int(os.environ.get("PB_B", "32")), int(sys.argv[1]) if len(sys.argv) > 1 else 400
g = torch.Generator().manual_seed(3)
gates0 = (torch.randn(T, B, 2, 4 * H, generator=g) * 0.5).to(dev)
wf = (torch.randn(2, 4 * H, H, generator=g) / np.sqrt(H)).to(dev)
w = (torch.randn(2, H, 4 * H, generator=g) / np.sqrt(H)).to(dev)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
y = torch.empty(T, B, 2 * H, device=dev); c = torch.empty(T, B, 2 * H, device=dev)
gact = (torch.rand(T, B, 2, 4 * H, generator=g) * 0.8 + 0.1).to(dev)
dy = (torch.randn(T, B, 2 * H, generator=g) * 0.01).to(dev); cc = torch.randn(T, B, 2 * H, generator=g).to(dev)
yy = torch.tanh(torch.randn(T, B, 2 * H, generator=g)).to(dev); dw = torch.zeros(2, 4 * H, H, device=dev)
xch, ctrl = hb.persist_scratch(dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = hb.stream()
def timeit(fn, n=3):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); rc = fn(); e1.record(); torch.cuda.synchronize(); assert rc == 0, rc
        best = min(best, e0.elapsed_time(e1) * 1e3 / T)
    return best
ARITHS = [hb._arith_code(a) for a in os.environ.get("PB_ARITH", "bf16x6,bf16x3").split(",")]
paths = [hb.LIB_PATH] + sorted(glob.glob(ROOT + '/scratchlibs/lib_*.so'))
libs = {p: ctypes.CDLL(p) for p in paths}
best = {(p, a): [1e9, 1e9, 1e9] for p in paths for a in ARITHS}
for rep in range(4):                 # interleaved repetitions: clocks / placement drift between runs
    for path in paths:
      for ar in ARITHS:
        l = libs[path]
        ga = gates0.clone()
        tf = timeit(lambda: l.asr_lstm_seq_fwd_persist(T, B, B, H, 2, P(ga), P(wf), P(lens), None, None, None, P(y), P(c), P(xch), P(ctrl), ar, st))
        gb = gact.clone()
        tb = timeit(lambda: l.asr_lstm_seq_bwd_persist(T, B, B, H, 2, P(gb), P(w), P(lens), None, None, None, P(dy), P(cc), P(yy), P(dw), None, P(xch), P(ctrl), ar, st))
        gb = gact.clone()
        tb0 = timeit(lambda: l.asr_lstm_seq_bwd_persist(T, B, B, H, 2, P(gb), P(w), P(lens), None, None, None, P(dy), P(cc), None, None, None, P(xch), P(ctrl), ar, st))
        best[(path, ar)] = [min(a_, b_) for a_, b_ in zip(best[(path, ar)], (tf, tb, tb0))]
        if int(ctrl[0].item()) != 0: print('ABORT in', os.path.basename(path), 'code', int(ctrl[1].item()), 'rep', rep, flush=True); ctrl[:2].zero_()
for path in paths:
    for ar in ARITHS:
        tf, tb, tb0 = best[(path, ar)]
        print('%-26s %-7s fwd %.2f us/step | bwd %.2f us/step (no dW: %.2f)' % (os.path.basename(path), hb.arith_name(ar), tf, tb, tb0), flush=True)

# a variant must compute what the shipped library computes (forward outputs; the hand-off's tag bit costs <= 1 ulp)
if len(paths) > 1:
    ref = {}
    for path in paths:
        ga = gates0.clone(); y.zero_(); c.zero_()
        assert libs[path].asr_lstm_seq_fwd_persist(T, B, B, H, 2, P(ga), P(wf), P(lens), None, None, None, P(y), P(c), P(xch), P(ctrl), ARITHS[0], st) == 0
        torch.cuda.synchronize()
        gb = gact.clone()
        assert libs[path].asr_lstm_seq_bwd_persist(T, B, B, H, 2, P(gb), P(w), P(lens), None, None, None, P(dy), P(cc), None, None, None, P(xch), P(ctrl), ARITHS[0], st) == 0
        torch.cuda.synchronize()
        out = (y.clone(), c.clone(), ga, gb)
        if path == hb.LIB_PATH:
            ref = out
        else:
            print('%-26s vs shipped: forward max |dy| %.2e  |dc| %.2e  |dgates| %.2e; backward max |d(dG)| %.2e of %.2e' % (
                os.path.basename(path), float((out[0] - ref[0]).abs().max()), float((out[1] - ref[1]).abs().max()),
                float((out[2] - ref[2]).abs().max()), float((out[3] - ref[3]).abs().max()), float(ref[3].abs().max())), flush=True)

# the same recurrences on PACKED rows (every utterance T frames + 8 padding rows: rowbase / rowext of include/asr_hip.h)
l = libs[hb.LIB_PATH]
ext = T + 8
R = B * ext
base_h = (np.arange(B) * ext).astype(np.int32); ext_h = np.full(B, ext, dtype=np.int32)
rbase, rext = torch.from_numpy(base_h).to(dev), torch.from_numpy(ext_h).to(dev)
pg0 = (torch.randn(R, 2, 4 * H, generator=g) * 0.5).to(dev); pga = (torch.rand(R, 2, 4 * H, generator=g) * 0.8 + 0.1).to(dev)
py = torch.empty(R, 2 * H, device=dev); pc = torch.empty(R, 2 * H, device=dev)
pdy = (torch.randn(R, 2 * H, generator=g) * 0.01).to(dev); pcc = torch.randn(R, 2 * H, generator=g).to(dev)
for ar in ARITHS:
    bf = bb = 1e9
    for rep in range(4):
        ga = pg0.clone()
        bf = min(bf, timeit(lambda: l.asr_lstm_seq_fwd_persist(T, B, B, H, 2, P(ga), P(wf), P(lens), P(rbase), P(rext), None, P(py), P(pc), P(xch), P(ctrl), ar, st)))
        gb = pga.clone()
        bb = min(bb, timeit(lambda: l.asr_lstm_seq_bwd_persist(T, B, B, H, 2, P(gb), P(w), P(lens), P(rbase), P(rext), None, P(pdy), P(pcc), None, None, None, P(xch), P(ctrl), ar, st)))
    print('%-26s %-7s fwd %.2f us/step | bwd %.2f us/step   (packed rows)' % (os.path.basename(hb.LIB_PATH), hb.arith_name(ar), bf, bb), flush=True)
