"""Per-phase timing of the persistent decoder kernels from shader-clock stamps (library built with -DASR_DP_TRACE in
scratchlibs/lib_trace.so): mean cycles between marks over decoder steps 8..15 of workgroup (group 0, slice 0)."""
import ctypes, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd']
import torch, numpy as np
import hip_backend as hb, ops
dev = torch.device('cuda')
B, Tp, L, D, E, C, K, V = 32, 100, 101, 512, 128, 10, 100, 34
A = O = D; KX = D + O + E
g = torch.Generator().manual_seed(5)
rnd = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g) * sc).to(dev)
ws = ops._dec_workspace(B, Tp, A, D, O, E, C, K, L, True, dev, True)
s0 = 1.0 / np.sqrt(D)
ws["P"].copy_(rnd(B, Tp, A, sc=0.5)); ws["Q"].copy_(rnd(B, Tp, O, sc=0.5)); ws["wcat"].copy_(rnd(4 * D, KX, sc=s0))
ws["bcat"].copy_(rnd(4 * D, sc=s0)); ws["convw"].copy_(rnd(C, 2 * K + 1, sc=0.1)); ws["gvec"].copy_(rnd(A, sc=s0))
watt = rnd(A, C, sc=0.3); ws["wattT"].copy_(watt.t()); ws["w0"].fill_(1.0 / Tp)
ws["xmask"].copy_((torch.rand(L, B, O + E, generator=g) > 0.3).float().to(dev) / 0.7)
ws["X"].zero_(); ws["X"][:L, :, D + O:] = rnd(L, B, E, sc=0.5)
ws["Xd"].zero_(); ws["Xd"][:L, :, D + O:] = ws["X"][:L, :, D + O:] * ws["xmask"][:, :, O:]
wdec = rnd(A, D, sc=s0)
d = dict(B=B, Tp=Tp, A=A, D=D, O=O, E=E, C=C, K=K, L=L, KX=KX, scaling=2.0, bo=rnd(O, sc=s0), wdec=wdec, watt=watt)
d.update({k: ws[k] for k in ("P", "Q", "wcat", "bcat", "convw", "gvec", "wattT", "w0", "xmask", "X", "Xd", "gates", "cstate",
                             "Dproj", "fconv", "S", "energy", "ws")})
for k in ("G", "dwext", "dP", "dcell", "dgvec_part", "dwatt_part", "dconv_part", "dws"):
    ws[k].zero_()
ws["G"][1:, :, :D + O] = rnd(L, B, D + O, sc=0.01)
ws["wcatT"].copy_(ws["wcat"].t()); ws["wdecT"].copy_(wdec.t())
w = dict(ws); w["dws"] = None
fs = ops._dec_fwd_struct(d, 0, B); bs = ops._dec_bwd_struct(d, w, 0, B)
xch, ctrl = hb.persist_scratch(dev, trace=True)
st = hb.stream()
l = ctypes.CDLL(ROOT + '/scratchlibs/lib_trace.so')
l.asr_dec_seq_fwd_persist.argtypes = [ctypes.POINTER(hb.DecFwd), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
l.asr_dec_seq_bwd_persist.argtypes = [ctypes.POINTER(hb.DecBwd), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
X, C_ = ctypes.c_void_p(xch.data_ptr()), ctypes.c_void_p(ctrl.data_ptr())
def report(name, nmarks):
    t = ctrl[32:32 + 8 * 16 * 2].cpu().numpy().view(np.int64).reshape(8, 16)[:, :nmarks]
    dl = np.diff(t, axis=1).mean(0)
    step = (t[1:, 0] - t[:-1, 0]).mean()
    print(name, 'cycles/step %.0f' % step, ' deltas:', ' '.join('%d:%.0f' % (i + 1, x) for i, x in enumerate(dl)))
for _ in range(2):
    ctrl.zero_()
    assert l.asr_dec_seq_fwd_persist(ctypes.byref(fs), X, C_, st) == 0
    torch.cuda.synchronize(); report('fwd', 11)
for _ in range(2):
    ctrl.zero_()
    assert l.asr_dec_seq_bwd_persist(ctypes.byref(bs), hb.ptr(ws["Mf"]), X, C_, st) == 0
    torch.cuda.synchronize(); report('bwd', 10)
    t = ctrl[32:32 + 8 * 16 * 2].cpu().numpy().view(np.int64).reshape(8, 16)
    print('   (e) split: df+barrier %.0f | dwext %.0f | dconv %.0f | prefetchA %.0f' % ((t[:, 10] - t[:, 4]).mean(), (t[:, 11] - t[:, 10]).mean(), (t[:, 12] - t[:, 11]).mean(), (t[:, 5] - t[:, 12]).mean()))
