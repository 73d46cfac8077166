"""Per-CU imbalance of the persistent LSTM backward kernel (library built with -DASR_LP_TRACE2 in
scratchlibs/lib_lptrace2.so): for every CU (slice) of group 0, how long wave 0 waits in the hand-off poll."""
import ctypes, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd']
import torch, numpy as np
import hip_backend as hb
dev = torch.device('cuda')
H, B, T = 512, 32, 64
g = torch.Generator().manual_seed(3)
w = (torch.randn(2, H, 4 * H, generator=g) / np.sqrt(H)).to(dev)
wf = (torch.randn(2, 4 * H, H, generator=g) / np.sqrt(H)).to(dev)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
gact = (torch.rand(T, B, 2, 4 * H, generator=g) * 0.8 + 0.1).to(dev)
gates0 = (torch.randn(T, B, 2, 4 * H, generator=g) * 0.5).to(dev)
dy = (torch.randn(T, B, 2 * H, generator=g) * 0.01).to(dev); cc = torch.randn(T, B, 2 * H, generator=g).to(dev)
yy = torch.tanh(torch.randn(T, B, 2 * H, generator=g)).to(dev); dw = torch.zeros(2, 4 * H, H, device=dev)
y = torch.empty(T, B, 2 * H, device=dev); c = torch.empty(T, B, 2 * H, device=dev)
db = torch.zeros(2 * 4 * H, device=dev)
xch = torch.zeros(2 * 8 * 8 * 2048, dtype=torch.int64, device=dev)
ctrl = torch.zeros(4096, dtype=torch.int32, device=dev)        # the probe writes 32 x 8 x 2 stamps from word 0 on
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = hb.stream()
l = ctypes.CDLL(ROOT + '/scratchlibs/lib_lptrace2.so')
def show(name):
    t = ctrl[16:16 + 1024].cpu().numpy().view(np.int64).reshape(32, 8, 2)
    wait = (t[:, :, 1] - t[:, :, 0]).mean(1)
    top = (t[:, :, 0] - t[:, :, 0].min(0, keepdims=True)).mean(1)
    print(name, 'poll wait per CU (cycles):', ' '.join('%d' % x for x in wait))
    print(name, 'arrival at the top of a step relative to the earliest CU:', ' '.join('%d' % x for x in top))
for _ in range(2):
    gb = gact.clone()
    assert l.asr_lstm_seq_bwd_persist(T, B, B, H, 2, P(gb), P(w), P(lens), None, None, None, P(dy), P(cc), P(yy), P(dw), P(db), P(xch), P(ctrl), hb.current_arith(), st) == 0
    torch.cuda.synchronize()
show('bwd')
for _ in range(2):
    ga = gates0.clone()
    assert l.asr_lstm_seq_fwd_persist(T, B, B, H, 2, P(ga), P(wf), P(lens), None, None, None, P(y), P(c), P(xch), P(ctrl), hb.current_arith(), st) == 0
    torch.cuda.synchronize()
show('fwd')
