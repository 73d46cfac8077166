"""What happens to a persistent XCD-local kernel when a foreign kernel is already resident on some CUs (the situation of an
RCCL collective that waits for a late rank while the next LSTM / decoder kernel is launched)?  A spinner (tools/micro/
spinner.hip -> scratchlibs/spinner.so) holds `wgs` workgroups for ~2 ms on a side stream; the persistent LSTM kernels are
launched behind it on the main stream.  Reported: kernel time alone / with the spinner resident, and the abort latch.
    hipcc --offload-arch=gfx950 -O3 -fPIC -shared -o scratchlibs/spinner.so tools/micro/spinner.hip
    python3 tools/coresident_probe.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd']
import torch, numpy as np, hip_backend as hb
dev = torch.device('cuda'); lib = hb.load()
sp = ctypes.CDLL(ROOT + '/scratchlibs/spinner.so')
H, B, T = 512, 32, 400
AR = hb.ARITH_BF16X6
g = torch.Generator().manual_seed(3)
gates0 = (torch.randn(T, B, 2, 4 * H, generator=g) * 0.5).to(dev)
wf = (torch.randn(2, 4 * H, H, generator=g) / np.sqrt(H)).to(dev)
w = (torch.randn(2, H, 4 * H, generator=g) / np.sqrt(H)).to(dev)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
y = torch.empty(T, B, 2 * H, device=dev); c = torch.empty(T, B, 2 * H, device=dev)
gact = (torch.rand(T, B, 2, 4 * H, generator=g) * 0.8 + 0.1).to(dev)
dy = (torch.randn(T, B, 2 * H, generator=g) * 0.01).to(dev); cc = torch.randn(T, B, 2 * H, generator=g).to(dev)
xch, ctrl = hb.persist_scratch(dev)
sink = torch.zeros(4, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
main = torch.cuda.current_stream(); side = torch.cuda.Stream()
def fwd():
    ga = gates0.clone()
    return lambda: lib.asr_lstm_seq_fwd_persist(T, B, B, H, 2, P(ga), P(wf), P(lens), None, None, None, P(y), P(c), P(xch), P(ctrl), AR, ctypes.c_void_p(main.cuda_stream))
def bwd():
    gb = gact.clone()
    return lambda: lib.asr_lstm_seq_bwd_persist(T, B, B, H, 2, P(gb), P(w), P(lens), None, None, None, P(dy), P(cc), None, None, None, P(xch), P(ctrl), AR, ctypes.c_void_p(main.cuda_stream))
def timed(run, spin=None):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    if spin is not None:
        wgs, thr, lds, regs, cyc = spin
        rc = sp.spin_launch(wgs, thr, lds, regs, ctypes.c_longlong(cyc), P(sink), ctypes.c_void_p(side.cuda_stream)); assert rc == 0, rc
    e0.record(main); rc = run(); e1.record(main); assert rc == 0, rc
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3
for name, mk in (('fwd', fwd), ('bwd', bwd)):
    run = mk(); timed(run)
    base = min(timed(mk()) for _ in range(3))
    print('%s alone: %.0f us (%.2f us/step)' % (name, base, base / T), flush=True)
    for spin in ((16, 256, 4096, 64, 4_000_000), (32, 512, 16384, 100, 4_000_000), (64, 256, 65536, 64, 4_000_000), (16, 512, 100000, 100, 4_000_000),
                 (256, 256, 4096, 64, 4_000_000)):
        t = timed(mk(), spin)
        ab = int(ctrl[0].item())
        print('  with %3d x %3d-thread spinner WGs (%6d B LDS, ~%3d VGPRs) resident ~1.7 ms: %.0f us (%.2f us/step)%s' % (
            spin[0], spin[1], spin[2], spin[3], t, t / T, '  ABORT code %d' % int(ctrl[1].item()) if ab else ''), flush=True)
        if ab: ctrl[:2].zero_()
