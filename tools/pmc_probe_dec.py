"""Runs only the persistent decoder kernels (forward, then att_m + backward) once at a given shape so that a rocprofv3
--pmc pass stays short; default = cfg-5 (B=8, T'=200, L+1=201: the T' <= 256 geometry, 2 utterances per XCD group):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -o f -- python3 tools/pmc_probe_dec.py [B Tp L]
and again with WRITE_SIZE.  tools/pmc_summary_dec.py turns the two csv files into bytes per decoder step and GB/s."""
import ctypes, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd']
import torch, numpy as np
import hip_backend as hb, ops
dev = torch.device('cuda')
B, Tp, L = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (8, 200, 201)
D, E, C, K, V = 512, 128, 10, 100, 34
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
ws["zbuf"].zero_()
ws["G"][1:, :, :D + O] = rnd(L, B, D + O, sc=0.01)
ws["wcatT"].copy_(ws["wcat"].t()); ws["wdecT"].copy_(wdec.t())
w = dict(ws); w["dws"] = None
fs = ops._dec_fwd_struct(d, 0, B); bs = ops._dec_bwd_struct(d, w, 0, B)
xch, ctrl = hb.persist_scratch(dev)
lib = hb.load()
X_, C_ = ctypes.c_void_p(xch.data_ptr()), ctypes.c_void_p(ctrl.data_ptr())
hb.check(lib.asr_dec_seq_fwd_persist(ctypes.byref(fs), X_, C_, hb.stream()), 'dec fwd persist')
hb.check(lib.asr_dec_seq_bwd_persist(ctypes.byref(bs), hb.ptr(ws["Mf"]), X_, C_, hb.stream()), 'dec bwd persist')
torch.cuda.synchronize()
print('done B=%d Tp=%d L=%d' % (B, Tp, L), 'aborted' if hb.persist_aborted(dev) else '')
