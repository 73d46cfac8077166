"""What the persistent decoder BACKWARD's step is made of, by removal (cfg-2 shape: B = 32, T' = 100, L = 101, D = A = O = 512):
the shipped library next to measurement builds with one phase reduced to a single trip (csrc/dec_persist.hip: ASR_DP_ABL bits
128 conv backward, 256 score backward, 512 dX product, 1024 dz product, 2048 no global stores of dgates / dD), built on the box:
    bash tools/mkvar_dec.sh libd_abl128 -DASR_DP_ABL=128 libd_abl256 -DASR_DP_ABL=256 ... (tools/mkvar_dec.sh)
    python3 tools/dec_bwd_ablate.py
Numbers of an ablated build are wrong by design; only its time means something: step(shipped) - step(without X) = what X
costs ON the chain (a phase that another one hides behind costs nothing when removed)."""
import ctypes, glob, os, sys
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
ws["wcatT"].copy_(ws["wcat"].t()); ws["wdecT"].copy_(wdec.t())
w = dict(ws); w["dws"] = None
fs = ops._dec_fwd_struct(d, 0, B); bs = ops._dec_bwd_struct(d, w, 0, B)
xch, ctrl = hb.persist_scratch(dev)
st = hb.stream()
X, C_ = ctypes.c_void_p(xch.data_ptr()), ctypes.c_void_p(ctrl.data_ptr())
base = ctypes.CDLL(hb.LIB_PATH)
base.asr_dec_seq_fwd_persist.argtypes = [ctypes.POINTER(hb.DecFwd), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
assert base.asr_dec_seq_fwd_persist(ctypes.byref(fs), X, C_, st) == 0          # the forward data the backward reads

names = {0: "shipped", 128: "conv backward: one trip of each Toeplitz product", 256: "no score backward", 512: "dX product: one quad",
         1024: "dz product: one pair", 2048: "no global stores (dgates, dD)", 1920: "no conv, scores, dX, dz products (128+256+512+1024)",
         3968: "... and no stores: polls, barriers, pointwise, prefetches"}
rows = []
for path in [hb.LIB_PATH] + sorted(glob.glob(ROOT + '/scratchlibs/libd_abl*.so')):
    l = ctypes.CDLL(path)
    l.asr_dec_seq_bwd_persist.argtypes = [ctypes.POINTER(hb.DecBwd), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    best = 1e9
    for rep in range(4):
        for k in ("G", "dwext", "dP", "dcell", "dgvec_part", "dwatt_part", "dconv_part"):
            ws[k].zero_()
        ws["G"][1:, :, :D + O] = 0.01
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); rc = l.asr_dec_seq_bwd_persist(ctypes.byref(bs), hb.ptr(ws["Mf"]), X, C_, st); e1.record(); torch.cuda.synchronize()
        assert rc == 0, rc
        if rep: best = min(best, e0.elapsed_time(e1) * 1e3 / L)
    bit = 0 if path == hb.LIB_PATH else int(os.path.basename(path)[len("libd_abl"):-3])
    rows.append((bit, best, int(ctrl[0].item())))
shipped = rows[0][1]
for bit, us, ab in rows:
    print("%-62s %6.2f us per decoder step  (%+.2f)%s" % (names.get(bit, "ASR_DP_ABL=%d" % bit), us, us - shipped, "  ABORT" if ab else ""))
    if ab: ctrl[:2].zero_()
