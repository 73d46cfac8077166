"""Diagnostic: persistent decoder backward in the T' <= 256 geometry vs the per-step kernels (where do they differ?)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd']
import numpy as np, torch
import __graft_entry__ as entry
entry.build()
import hip_backend as hb
if os.environ.get('DBGLIB'):
    hb.LIB_PATH = ROOT + '/scratchlibs/' + os.environ['DBGLIB']; hb._lib = None; hb.load()
import ops
dev = torch.device('cuda')
dim, B, Tp, L, drop = 512, int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 200, int(os.environ.get('LL', '4')), True
g = torch.Generator().manual_seed(13 + B + Tp)
D = A = O = dim
E, C, K, V = 128, int(os.environ.get("CC", "10")), int(os.environ.get("KK", "100")), 34
sc0 = 1.0 / np.sqrt(D)
rnd = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g) * sc).to(dev)
base = dict(P=rnd(B, Tp, A, sc=float(os.environ.get("P_SC", "0.5"))), Q=rnd(B, Tp, O, sc=0.5), emb_w=rnd(V, E, sc=0.5), w_ih=rnd(4 * D, E + O, sc=sc0),
            w_hh=rnd(4 * D, D, sc=sc0), b_ih=rnd(4 * D, sc=sc0), b_hh=rnd(4 * D, sc=sc0), wdec=rnd(A, D, sc=sc0 * float(os.environ.get("WDEC_SC", "1"))),
            convw=rnd(C, 1, 1, 2 * K + 1, sc=float(os.environ.get("CONV_SC", "0.1"))), watt=rnd(A, C, sc=float(os.environ.get("WATT_SC", "0.3"))), gvec=rnd(1, A, sc=sc0), bo=rnd(O, sc=sc0),
            w_out=rnd(V, D + O, sc=sc0), b_out=rnd(V, sc=sc0))
lens = torch.randint(max(1, Tp // 2), Tp + 1, (B,), generator=g)
w0 = torch.zeros(B, Tp)
for b in range(B):
    w0[b, :lens[b]] = 1.0 / float(lens[b])
w0 = w0.to(dev)
tokens = torch.randint(0, V, (B, L), generator=g).to(dev)
xmask = ((torch.rand(L, B, O + E, generator=g) > 0.3).float() / 0.7).to(dev) if drop else None
dlog = rnd(L, B, V); dws = rnd(L, B, Tp, sc=float(os.environ.get("DWS_SC", "0.1")))
names = list(base.keys())
def run(persist, persist_bwd):
    hb.USE_PERSIST_DEC, hb.USE_PERSIST_DEC_BWD = persist, persist_bwd
    par = {k: v.clone().requires_grad_(True) for k, v in base.items()}
    opts = dict(L=L, tokens=tokens, tf_flags=None, smooth=False, sample=False, scaling=2.0, xmask=xmask, bos=1)
    hb.LAUNCHES.clear()
    logits, ws, _ = ops.decoder_sequence(par["P"], par["Q"], par["emb_w"], par["w_ih"], par["w_hh"], par["b_ih"], par["b_hh"],
                                         par["wdec"], par["convw"], par["watt"], par["gvec"], par["bo"], par["w_out"], par["b_out"], w0, opts)
    ((logits * dlog).sum() + (ws * dws).sum()).backward()
    torch.cuda.synchronize()
    print(dict(hb.LAUNCHES), 'aborted', hb.persist_aborted(dev))
    return {k: par[k].grad.detach() for k in names}
gr = run(False, False)
gp = run(False, True)
for k in names:
    d = (gp[k] - gr[k]).abs()
    scale = float(gr[k].abs().max()) + 1e-12
    idx = np.unravel_index(int(d.argmax()), d.shape)
    print('%-8s rel err %.2e at %s (got %.4e want %.4e)' % (k, float(d.max()) / scale, idx, float(gp[k][idx]), float(gr[k][idx])))
dP = (gp["P"] - gr["P"]).abs()
print('dP err by row:', [float(x) for x in dP.amax(dim=(1, 2))])
print('dP err by frame (top):', sorted([(float(v), i) for i, v in enumerate(dP.amax(dim=(0, 2)))], reverse=True)[:8])
print('dP err by col block of 16:', [round(float(x), 6) for x in dP.amax(dim=(0, 1)).view(-1, 16).amax(1)][:8])

# float64 autograd restatement of the same sequence (teacher forced): which of the two backward paths is closer?
import torch.nn.functional as F
def ref64():
    par = {k: v.detach().double().cpu().requires_grad_(True) for k, v in base.items()}
    P, Q, emb_w = par["P"], par["Q"], par["emb_w"]
    w_prev = w0.double().cpu()
    z = torch.zeros(B, D, dtype=torch.float64); c = torch.zeros(B, D, dtype=torch.float64); ctx = torch.zeros(B, O, dtype=torch.float64)
    xm = xmask.double().cpu() if xmask is not None else None
    tok = tokens.cpu()
    logits, wsl = [], []
    for s in range(L):
        emb = emb_w[tok[:, s]]
        cin, ein = (ctx * xm[s][:, :O], emb * xm[s][:, O:]) if xm is not None else (ctx, emb)
        gates = torch.cat([ein, cin], 1) @ par["w_ih"].t() + par["b_ih"] + z @ par["w_hh"].t() + par["b_hh"]
        gi, gf, gg, go = gates.split(D, 1)
        c = torch.sigmoid(gf) * c + torch.sigmoid(gi) * torch.tanh(gg)
        z = torch.sigmoid(go) * torch.tanh(c)
        f = F.conv1d(w_prev.unsqueeze(1), par["convw"].reshape(C, 1, -1), padding=K)            # [B,C,Tp]
        f.retain_grad()
        S_ = torch.tanh(P + (z @ par["wdec"].t()).unsqueeze(1) + f.transpose(1, 2) @ par["watt"].t())
        e = (S_ @ par["gvec"].t()).squeeze(2)
        e.retain_grad()
        ref64.last = (f, S_, e)
        w = torch.softmax(2.0 * e, dim=1)
        ctx = torch.bmm(w.unsqueeze(1), Q).squeeze(1) + par["bo"]
        logits.append(torch.cat([z, ctx], 1) @ par["w_out"].t() + par["b_out"]); wsl.append(w)
        w_prev = w
    ((torch.stack(logits) * dlog.double().cpu()).sum() + (torch.stack(wsl) * dws.double().cpu()).sum()).backward()
    return {k: par[k].grad for k in names}
g64 = ref64()
for k in ("P", "wdec", "convw", "watt", "gvec", "Q", "w_ih"):
    sc = float(g64[k].abs().max())
    print('%-6s vs float64: per-step %.2e   persistent %.2e' % (k, float((gr[k].double().cpu() - g64[k]).abs().max()) / sc,
                                                                   float((gp[k].double().cpu() - g64[k]).abs().max()) / sc))
d64 = (gp["P"].double().cpu() - g64["P"]).abs()
per_frame = d64.amax(dim=(0, 2)) / g64["P"].abs().amax(dim=(0, 2)).clamp_min(1e-12)
print('P rel err per frame:', ' '.join('%d:%.0e' % (i, float(v)) for i, v in enumerate(per_frame)))
per_row = d64.amax(dim=(1, 2)) / g64["P"].abs().amax(dim=(1, 2))
print('P rel err per row:', [float('%.1e' % float(v)) for v in per_row])

if L == 2:
    row = 0
    e = (gp["P"].double().cpu() - g64["P"])[row].abs().amax(1)          # per frame
    ref = g64["P"][row].abs().amax(1)
    print('len', int(lens[row]), 'row0 per-frame abs err (x1e6):', ' '.join('%d:%.0f' % (i, float(v) * 1e6) for i, v in enumerate(e)))
    print('row0 per-frame |grad| (x1e6):', ' '.join('%d:%.0f' % (i, float(v) * 1e6) for i, v in enumerate(ref)))

if os.environ.get('DBGLIB'):
    key = [k for k in ops._POOL.free if k[0] == "dec"][0]
    dbg = ops._POOL.free[key][-1]["dfpart"].reshape(-1)[:3 * B * C * Tp].view(3, B, C, Tp).double().cpu()
    f64, S64, e64 = ref64.last
    U = base["watt"].double().cpu(); gvv = base["gvec"].double().cpu().view(-1)
    M64 = torch.einsum('ac,a,bta->bct', U, gvv, S64.detach() ** 2)
    print('df  max rel err %.2e' % float((dbg[0] - f64.grad).abs().max() / f64.grad.abs().max()))
    print('M   max rel err %.2e' % float((dbg[1] - M64).abs().max() / M64.abs().max()))
    print('des max rel err %.2e' % float((dbg[2] - e64.grad.unsqueeze(1).expand(-1, C, -1)).abs().max() / e64.grad.abs().max()))
    dM = (dbg[1] - M64).abs()
    print('M err per channel', [float('%.1e' % float(v)) for v in dM.amax(dim=(0, 2))], 'per row', [float('%.1e' % float(v)) for v in dM.amax(dim=(1, 2))])
    print('M err per frame (row 0, ch 0):', ' '.join('%d:%.0e' % (i, float(v)) for i, v in enumerate(dM[0, 0])))
if os.environ.get('DBGLIB'):
    print('dbg des row0[:6]', dbg[2][0, 0, :6].tolist(), '\nref des row0[:6]', e64.grad[0, :6].tolist())
    print('dbg M row0 ch0[:6]', dbg[1][0, 0, :6].tolist(), '\nref M[:6]', M64[0, 0, :6].tolist())
    print('ratio des', (dbg[2][0, 0, :6] / e64.grad[0, :6]).tolist())

if os.environ.get('DBGLIB'):
    raw = ops._POOL.free[key][-1]["dfpart"].reshape(-1)
    dwx = raw[3 * B * C * Tp: 3 * B * C * Tp + B * Tp].view(B, Tp).double().cpu()
    Fw = base["convw"].double().cpu().view(C, 2 * K + 1)
    dfp = torch.zeros(B, C, Tp + 2 * K, dtype=torch.float64); dfp[:, :, K:K + Tp] = f64.grad
    ref = torch.zeros(B, Tp, dtype=torch.float64)
    for j in range(2 * K + 1):       # dw[t'] = sum_c sum_j F[c][j] df[c][t' - j + K]
        ref += (Fw[:, j].view(1, C, 1) * dfp[:, :, 2 * K - j: 2 * K - j + Tp]).sum(1)
    for b in range(B):
        ref[b, int(lens[b]) * 0 + Tp:] = 0
    err = (dwx - ref).abs()
    print('dwext max rel err %.2e' % float(err.max() / ref.abs().max()))
    print('dwext err per frame (row 0):', ' '.join('%d:%.0e' % (i, float(v)) for i, v in enumerate(err[0] / ref.abs().max())))
    print('dwext err per row:', [float('%.1e' % float(v)) for v in err.amax(1) / ref.abs().max()])
