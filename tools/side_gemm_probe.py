"""cfg-5 (B = 8: the persistent chains use four of the eight XCDs): do weight-gradient GEMMs on a SIDE stream run beside the
chain - with the EXISTING kernels?  Main stream: asr_lstm_seq_bwd_persist (B = 8, T = 800).  Side stream: the dW products of
the layer above ([4096, 512] = dG^T x over K rows, and the batched dW_hh), forced onto a given tile family.
Reported: chain alone, GEMMs alone, both at once (wall from the first launch to the last completion)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd']
import torch, numpy as np, hip_backend as hb
dev = torch.device('cuda'); lib = hb.load()
H, B, T = 512, int(os.environ.get("B", "8")), int(os.environ.get("T", "800"))
AR = hb.ARITH_BF16X6
g = torch.Generator().manual_seed(3)
w = (torch.randn(2, H, 4 * H, generator=g) / np.sqrt(H)).to(dev)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
gact = (torch.rand(T, B, 2, 4 * H, generator=g) * 0.8 + 0.1).to(dev)
dy = (torch.randn(T, B, 2 * H, generator=g) * 0.01).to(dev); cc = torch.randn(T, B, 2 * H, generator=g).to(dev)
xch, ctrl = hb.persist_scratch(dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
prio = os.environ.get("PRIO", "1") == "1"
main = torch.cuda.Stream(priority=-1) if prio else torch.cuda.current_stream()
side = torch.cuda.Stream(priority=0)
R = int(os.environ.get("R", "10496"))                 # rows of the layer above at cfg-5
# (the row-shifted dW_hh product reads ONE row behind R on each side: both operands hold R + 1 rows, as ops._lstm_workspace does)
dG = torch.randn(R + 1, 8 * H, device=dev); x = torch.randn(R, H, device=dev); yv = torch.randn(R + 1, 2 * H, device=dev)
dw_ih = torch.zeros(8 * H, H, device=dev); dw_hh = torch.zeros(2, 4 * H, H, device=dev)
def chain():
    gb = gact.clone()
    torch.cuda.synchronize()
    def run():
        with torch.cuda.stream(main):
            rc = lib.asr_lstm_seq_bwd_persist(T, B, B, H, 2, P(gb), P(w), P(lens), None, None, None, P(dy), P(cc), None, None, None, P(xch), P(ctrl), AR, ctypes.c_void_p(main.cuda_stream))
            assert rc == 0, rc
    return run
def gemms(mode):
    def run():
        with torch.cuda.stream(side), hb.arith(mode):
            hb.gemm(dG[:R], x, trans_a=True, out=dw_ih, accumulate=True)
            hb.gemm_batched(dG, yv, dw_hh, True, False, 4 * H, H, R, 8 * H, 2 * H, H, 2, 4 * H - 8 * H, 2 * H + H, 4 * H * H, accumulate=True, a_off=8 * H, b_off=0)
    return run
def wall(fns):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    base = torch.cuda.current_stream()
    e0.record(base)
    main.wait_event(e0); side.wait_event(e0)
    for f in fns: f()
    base.wait_stream(main); base.wait_stream(side)
    e1.record(base)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3
print("B = %d, T = %d, dW rows %d, main stream priority %s" % (B, T, R, "high" if prio else "default"))
for mode in ("bf16x6", "bf16x6+small", "bf16x6+narrow"):
    wall([chain()]); wall([gemms(mode)])
    tc = min(wall([chain()]) for _ in range(3)); tg = min(wall([gemms(mode)]) for _ in range(3))
    both = [wall([gemms(mode), chain()]) for _ in range(3)]
    both2 = [wall([chain(), gemms(mode)]) for _ in range(3)]
    ab = int(ctrl[0].item())
    print("%-14s chain alone %.0f us | GEMMs alone %.0f us | GEMMs launched first: both %s us | chain launched first: both %s us%s"
          % (mode, tc, tg, " ".join("%.0f" % v for v in both), " ".join("%.0f" % v for v in both2), "  ABORT" if ab else ""), flush=True)
    if ab: ctrl[:2].zero_()
