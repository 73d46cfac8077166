"""Store -> visible latency of one hand-off granule inside the persistent LSTM forward kernel, on the chip-wide 100 MHz
clock (library built with -DASR_LP_TRACE3 in scratchlibs/lib_lptrace3.so)."""
import ctypes, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd']
import torch, numpy as np
import hip_backend as hb
dev = torch.device('cuda')
H, B, T = 512, 32, 80
g = torch.Generator().manual_seed(3)
wf = (torch.randn(2, 4 * H, H, generator=g) / np.sqrt(H)).to(dev)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
gates0 = (torch.randn(T, B, 2, 4 * H, generator=g) * 0.5).to(dev)
y = torch.empty(T, B, 2 * H, device=dev); c = torch.empty(T, B, 2 * H, device=dev)
xch = torch.zeros(2 * 8 * 8 * 2048, dtype=torch.int64, device=dev)
ctrl = torch.zeros(4096, dtype=torch.int32, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
l = ctypes.CDLL(ROOT + '/scratchlibs/lib_lptrace3.so')
for rep in range(3):
    ga = gates0.clone(); ctrl.zero_()
    assert l.asr_lstm_seq_fwd_persist(T, B, B, H, 2, P(ga), P(wf), P(lens), None, None, None, P(y), P(c), P(xch), P(ctrl), hb.current_arith(), hb.stream()) == 0
    torch.cuda.synchronize()
    t = ctrl[48:48 + 64 * 4].cpu().numpy().view(np.int64).reshape(64, 2)[8:60]
    ok = (t[:, 0] > 0) & (t[:, 1] > 0)
    d = (t[ok, 1] - t[ok, 0]) * 10.0
    print('publish -> first seen valid by a spinning wave of another CU: mean %.0f ns, median %.0f, min %.0f, max %.0f (n=%d); step %.0f ns'
          % (d.mean(), np.median(d), d.min(), d.max(), ok.sum(), np.diff(t[ok, 0]).mean() * 10.0))
