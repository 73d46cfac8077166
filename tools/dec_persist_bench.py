"""Times the persistent decoder forward kernel (cfg-2 shape: B=32, T'=100, L=101, D=A=O=512, E=128, dropout) against
the per-step launch chain, for the shipped library and any measurement variants under scratchlibs/."""
import ctypes, sys, os, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd']
import torch, numpy as np
import hip_backend as hb, ops
dev = torch.device('cuda')
B, Tp, L, D, E, C, K, V = 32, 100, 101, 512, 128, 10, 100, 34
A = O = D
KX = D + O + E
g = torch.Generator().manual_seed(5)
rnd = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g) * sc).to(dev)
ws = ops._dec_workspace(B, Tp, A, D, O, E, C, K, L, True, dev, False)
s0 = 1.0 / np.sqrt(D)
ws["P"].copy_(rnd(B, Tp, A, sc=0.5)); ws["Q"].copy_(rnd(B, Tp, O, sc=0.5)); ws["wcat"].copy_(rnd(4 * D, KX, sc=s0))
ws["bcat"].copy_(rnd(4 * D, sc=s0)); ws["convw"].copy_(rnd(C, 2 * K + 1, sc=0.1)); ws["gvec"].copy_(rnd(A, sc=s0))
watt = rnd(A, C, sc=0.3); ws["wattT"].copy_(watt.t()); ws["w0"].fill_(1.0 / Tp)
ws["xmask"].copy_((torch.rand(L, B, O + E, generator=g) > 0.3).float().to(dev) / 0.7)
ws["X"].zero_(); ws["X"][:L, :, D + O:] = rnd(L, B, E, sc=0.5)
ws["Xd"].zero_(); ws["Xd"][:L, :, D + O:] = ws["X"][:L, :, D + O:] * ws["xmask"][:, :, O:]
d = dict(B=B, Tp=Tp, A=A, D=D, O=O, E=E, C=C, K=K, L=L, KX=KX, scaling=2.0, bo=rnd(O, sc=s0), wdec=rnd(A, D, sc=s0), watt=watt)
d.update({k: ws[k] for k in ("P", "Q", "wcat", "bcat", "convw", "gvec", "wattT", "w0", "xmask", "X", "Xd", "gates", "cstate",
                             "Dproj", "fconv", "S", "energy", "ws")})
fs = ops._dec_fwd_struct(d, 0, B)
xch, ctrl = hb.persist_scratch(dev)
st = hb.stream()
def timeit(fn, n=3):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); rc = fn(); e1.record(); torch.cuda.synchronize(); assert rc == 0, rc
        best = min(best, e0.elapsed_time(e1) * 1e3 / L)
    return best
libs = [hb.LIB_PATH] + sorted(glob.glob(ROOT + '/scratchlibs/lib_*.so'))
ref = None
for path in libs:
    l = ctypes.CDLL(path)
    for f in ("asr_dec_seq_fwd_persist", "asr_dec_seq_fwd"):
        getattr(l, f).restype = ctypes.c_int
    l.asr_dec_seq_fwd_persist.argtypes = [ctypes.POINTER(hb.DecFwd), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    l.asr_dec_seq_fwd.argtypes = [ctypes.POINTER(hb.DecFwd), ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    if ref is None:
        te = timeit(lambda: l.asr_dec_seq_fwd(ctypes.byref(fs), 0, L, None, st))
        ref = ws["ws"].clone()
        print('per-step chain: %.2f us/step' % te, flush=True)
    tp = timeit(lambda: l.asr_dec_seq_fwd_persist(ctypes.byref(fs), ctypes.c_void_p(xch.data_ptr()), ctypes.c_void_p(ctrl.data_ptr()), st))
    print('%-28s persistent %.2f us/step  abort %d err %d  max |dw| %.2e' % (os.path.basename(path), tp, int(ctrl[0].item()),
          int(ctrl[1].item()), float((ws["ws"] - ref).abs().max())), flush=True)
