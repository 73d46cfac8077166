"""Who waits for whom in the persistent decoder kernels: stamps of EVERY slice (CU) of group 0 on the chip-wide 100 MHz
clock, decoder steps 8..15 (library built with -DASR_DP_TRACE2: bash tools/mkvar.sh lib_trace2 -DASR_DP_TRACE2).
Forward marks: 0 step top, 1 ctx poll done, 2 cell product done, 3 pointwise done (z published), 4 conv features done,
5 z / f poll done, 6 W_dec z done, 7 scores done (partial energies published), 8 energy poll done, 9 softmax done,
10 context done (ctx published).  For each of the three hand-offs on the critical path: when the producers publish
(first / median / last slice, and which slice is last), when the consumers come out of their poll, and the distance
between the LAST publication and the LAST poll exit (what the exchange itself costs) - per step, in microseconds."""
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
l = ctypes.CDLL(ROOT + '/scratchlibs/' + (sys.argv[1] if len(sys.argv) > 1 else 'lib_trace2.so'))
l.asr_dec_seq_fwd_persist.argtypes = [ctypes.POINTER(hb.DecFwd), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
l.asr_dec_seq_bwd_persist.argtypes = [ctypes.POINTER(hb.DecBwd), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
X, C_ = ctypes.c_void_p(xch.data_ptr()), ctypes.c_void_p(ctrl.data_ptr())


def stamps():
    t = ctrl[32:32 + 32 * 8 * 16 * 2].cpu().numpy().view(np.int64).reshape(32, 8, 16).astype(np.float64) * 0.01   # us
    return t - t[:, 0:1, 0:1].min()


def handoff(t, name, pub, done, steps):
    """pub: mark after which a slice has published; done: mark after the consumers' poll.  The poll of step s waits for the
    publications of step s (same step) or s - 1 (ctx -> next step's cell): `steps` = (producer step, consumer step) pairs."""
    rows = []
    for ps, cs in steps:
        p, c = t[:, ps, pub], t[:, cs, done]
        rows.append((p.min(), np.median(p), p.max(), int(p.argmax()), c.min(), np.median(c), c.max(), int(c.argmax())))
    r = np.array([x[:3] + x[4:7] for x in rows])
    last_p = [x[3] for x in rows]; last_c = [x[7] for x in rows]
    print(' %-34s producers publish over %.2f us (first -> last slice; median -> last %.2f); last publication -> median poll exit %.2f us, -> last poll exit %.2f us'
          % (name, (r[:, 2] - r[:, 0]).mean(), (r[:, 2] - r[:, 1]).mean(), (r[:, 4] - r[:, 2]).mean(), (r[:, 5] - r[:, 2]).mean()))
    print('   last producer slice per step: %s; last consumer: %s' % (last_p, last_c))


for name, call, nm in (('forward', lambda: l.asr_dec_seq_fwd_persist(ctypes.byref(fs), X, C_, st), 11),
                       ('backward', lambda: l.asr_dec_seq_bwd_persist(ctypes.byref(bs), hb.ptr(ws["Mf"]), X, C_, st), 10)):
    for rep in range(2):
        ctrl.zero_()
        assert call() == 0
        torch.cuda.synchronize()
    t = stamps()
    step = (t[:, 1:, 0] - t[:, :-1, 0]).mean()
    print('== %s: %.2f us per decoder step (all 32 slices of group 0, steps 8..15)' % (name, step))
    ph = np.diff(t[:, :, :nm], axis=2)                      # [slice][step][phase]
    print(' phase durations, us: mean over slices and steps | slowest slice (mean over steps) | fastest slice')
    for k in range(nm - 1):
        per = ph[:, :, k].mean(1)
        print('   %2d -> %2d  %6.2f | %6.2f (slice %2d) | %6.2f (slice %2d)' % (k, k + 1, per.mean(), per.max(), per.argmax(), per.min(), per.argmin()))
    tail = (t[:, 1:, 0] - t[:, :-1, nm - 1]).mean()
    print('   %2d -> next top %6.2f' % (nm - 1, tail))
    if name == 'forward':
        handoff(t, 'ctx_{s-1} -> cell product (mark 10 -> 1)', 10, 1, [(s, s + 1) for s in range(7)])
        handoff(t, 'z_s -> W_dec z (mark 3 -> 5)', 3, 5, [(s, s) for s in range(8)])
        handoff(t, 'partial energies -> softmax (7 -> 8)', 7, 8, [(s, s) for s in range(8)])
    np.save(ROOT + '/gpurun_out/dec_trace2_%s.npy' % name, t)
