"""cfg-5's question: B = 8 puts the persistent LSTM chains on four of the eight XCDs.  Can a foreign kernel use the
other four while the chain runs - and does it matter whether its workgroups would FIT beside a persistent workgroup?  A spinner
whose workgroups leave at once on the chain's XCDs (tools/micro/spinner.hip: xcd_mask) holds the other XCDs for ~3 ms on a side
stream; the persistent kernels (B = 8, T = 400) are launched behind it on the main stream.
    hipcc --offload-arch=gfx950 -O3 -fPIC -shared -o scratchlibs/spinner.so tools/micro/spinner.hip
    python3 tools/idle_xcd_probe.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd']
import torch, numpy as np, hip_backend as hb
dev = torch.device('cuda'); lib = hb.load()
sp = ctypes.CDLL(ROOT + '/scratchlibs/spinner.so')
H, B, T = 512, int(os.environ.get("B", "8")), 400
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
# which XCDs does the chain use?  ctrl[16 + x] ... not exposed: read it off the role counters after a launch (ctrl words 0-7
# are zeroed per launch by the library; the tickets taken stay until the next launch)
def fwd():
    ga = gates0.clone()
    return lambda: lib.asr_lstm_seq_fwd_persist(T, B, B, H, 2, P(ga), P(wf), P(lens), None, None, None, P(y), P(c), P(xch), P(ctrl), AR, ctypes.c_void_p(main.cuda_stream))
def bwd():
    gb = gact.clone()
    return lambda: lib.asr_lstm_seq_bwd_persist(T, B, B, H, 2, P(gb), P(w), P(lens), None, None, None, P(dy), P(cc), None, None, None, P(xch), P(ctrl), AR, ctypes.c_void_p(main.cuda_stream))
def timed(run, spin=None):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    s0 = torch.cuda.Event(enable_timing=True); s1 = torch.cuda.Event(enable_timing=True)
    if spin is not None:
        wgs, thr, lds, regs, cyc, mask = spin
        s0.record(side)
        rc = sp.spin_launch_mask(wgs, thr, lds, regs, ctypes.c_longlong(cyc), P(sink), ctypes.c_uint(mask), ctypes.c_void_p(side.cuda_stream)); assert rc == 0, rc
        s1.record(side)
    e0.record(main); rc = run(); e1.record(main); assert rc == 0, rc
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3, (s0.elapsed_time(s1) * 1e3 if spin is not None else 0.0)
print("B = %d rows, T = %d, H = %d" % (B, T, H))
for name, mk in (('fwd', fwd), ('bwd', bwd)):
    run = mk(); timed(run)
    base = min(timed(mk())[0] for _ in range(3))
    print('%s alone: %.0f us (%.2f us/step)' % (name, base, base / T), flush=True)
    IDLE = int(os.environ.get("IDLE_MASK", "0xcc"), 0)          # XCDs 2, 3, 6, 7
    for label, spin in (("fits beside a persistent workgroup, idle XCDs only", (256, 256, 60000, 100, 7_000_000, IDLE)),
                        ("does NOT fit (100 KB LDS), idle XCDs only", (256, 256, 100000, 100, 7_000_000, IDLE)),
                        ("does NOT fit (100 KB LDS), 512 workgroups, idle XCDs only", (512, 256, 100000, 100, 7_000_000, IDLE)),
                        ("fits, ALL XCDs", (256, 256, 60000, 100, 7_000_000, 0xff)),
                        ("does NOT fit, ALL XCDs", (256, 256, 100000, 100, 7_000_000, 0xff))):
        t, ts = timed(mk(), spin)
        ab = int(ctrl[0].item())
        print('  spinner %-62s (itself %.0f us): chain %.0f us (%.2f us/step)%s' % (label, ts, t, t / T,
              '  ABORT code %d' % int(ctrl[1].item()) if ab else ''), flush=True)
        if ab: ctrl[:2].zero_()
