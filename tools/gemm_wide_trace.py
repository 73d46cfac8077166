"""Per-stage shader-clock stamps of the wide split-bf16 GEMM (library built with -DASR_GW_TRACE: tools/mkvar.sh,
scratchlibs/trace_gw.so): waves 0 and 4 of workgroup 0; marks: stage top, H1 issued, DMA of the next stage landed,
barrier passed, H2 issued.  Each mark costs about 190 cycles itself."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd']
import numpy as np, torch
import hip_backend as hb
hb.LIB_PATH = ROOT + '/scratchlibs/' + (sys.argv[1] if len(sys.argv) > 1 else 'trace_gw.so')
hb._lib = None
lib = hb.load()
dev = torch.device('cuda')
shapes = {'nt': (0, 1, 12800, 4096, 512), 'tn': (1, 0, 4096, 512, 12800), 'nn': (0, 0, 12800, 512, 4096)}
for name in (sys.argv[2:] or ['nt', 'tn', 'nn']):
    ta, tb, M, N, K = shapes[name]
    A = torch.randn((K, M) if ta else (M, K), device=dev); B = torch.randn((N, K) if tb else (K, N), device=dev)
    out = torch.empty(M, N, device=dev)
    for _ in range(2):
        hb.gemm(A, B, trans_a=bool(ta), trans_b=bool(tb), out=out)
    torch.cuda.synchronize()
    buf = np.zeros(2 * 64 * 8, dtype=np.uint64)
    assert lib.asr_gw_trace_read(ctypes.c_void_p(buf.ctypes.data)) == 0
    t = buf.reshape(2, 64, 8).astype(np.int64)
    print('==', name, M, N, K)
    for w in range(2):
        tt = t[w]
        n = int((tt[:, 0] > 0).sum())
        rows = tt[4:min(n, 40) - 1]
        per = np.diff(tt[4:min(n, 40), 0])
        d = np.diff(rows[:, :5], axis=1)
        print(' K half %d: %d stages traced; cycles per stage %.0f | H1 issue %.0f | vmcnt+lgkm wait %.0f | barrier %.0f | H2 issue %.0f | to next top %.0f' % (
            w, n, per.mean(), d[:, 0].mean(), d[:, 1].mean(), d[:, 2].mean(), d[:, 3].mean(), (tt[5:min(n, 40), 0] - rows[:, 4]).mean()))
    c = t[0, 63, :5]
    print(' wave 0 coarse (cycles): prologue %d | main loop %d | to exchange %d | exchange %d | output %d | total %d' % (
        c[1] - c[0], c[2] - c[1], 0, c[3] - c[2], c[4] - c[3], c[4] - c[0]))
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); hb.gemm(A, B, trans_a=bool(ta), trans_b=bool(tb), out=out); e1.record(); torch.cuda.synchronize()
    print(' launch (events, warm): %.0f us' % (e0.elapsed_time(e1) * 1e3))
