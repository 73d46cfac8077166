"""torch.autograd.Function wrappers over the C-ABI kernels (hip_backend).

One Function per *sequence-level* operator so that the sequential chains (encoder
recurrence, decoder loop) never bounce through Python/autograd per time step:

    linear            y = relu?(x W^T + b)                       -> asr_gemm_f32
    lstm_layer        whole (bi)LSTM layer, time-major            -> asr_gemm_f32 + asr_lstm_seq_*
    pyramid_concat    pair-concat (+dropout mask)                 -> asr_pyramid_concat_*
    decoder_sequence  all decoder steps incl. attention           -> asr_dec_* (+ GEMMs)

GPU only; see hip_backend for the no-fallback rule.
"""
import ctypes

import torch

import hip_backend as hb


def gate_perm(H, device):
    """Row permutation torch (gate-major i,f,g,o) -> gate-interleaved (unit*4+gate)."""
    return torch.arange(4 * H, device=device).view(4, H).t().reshape(-1)


def gate_unperm(H, device):
    return torch.arange(4 * H, device=device).view(H, 4).t().reshape(-1)


# --------------------------------------------------------------------------------------
class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        x2 = x.reshape(-1, x.shape[-1])
        y = hb.gemm(x2, weight, trans_b=True, bias=bias, relu=relu)
        ctx.save_for_backward(x2, weight, y if relu else None)
        ctx.relu = relu
        ctx.has_bias = bias is not None
        ctx.in_shape = x.shape
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, weight, y = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        if ctx.relu:
            dy2 = dy2 * (y > 0).to(dy2.dtype)
        elif not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dx = hb.gemm(dy2, weight).view(ctx.in_shape) if ctx.needs_input_grad[0] else None
        dw = hb.gemm(dy2, x2, trans_a=True)
        db = hb.colsum(dy2) if ctx.has_bias else None
        return dx, dw, db, None


def linear(x, weight, bias=None, relu=False):
    """nn.Linear (+ optional fused ReLU) on the f32 MFMA GEMM (model.py:93-94,144)."""
    return _Linear.apply(x, weight, bias, relu)


# --------------------------------------------------------------------------------------
class _LstmLayer(torch.autograd.Function):
    """One (bi)directional LSTM layer over a padded time-major batch (model.py:79-81).
    params: for each direction w_ih [4H,I], w_hh [4H,H], b_ih [4H], b_hh [4H] (torch layout)."""

    @staticmethod
    def forward(ctx, x, lens, ndir, *params):
        T, B, I = x.shape
        H = params[1].shape[1]
        dev = x.device
        perm = gate_perm(H, dev)
        x2 = x.reshape(T * B, I)
        w_ih = torch.cat([params[4 * d][perm] for d in range(ndir)], 0)                       # [ndir*4H, I]
        bias = torch.cat([(params[4 * d + 2] + params[4 * d + 3])[perm] for d in range(ndir)], 0)
        w_hh = torch.stack([params[4 * d + 1][perm] for d in range(ndir)], 0).contiguous()     # [ndir,4H,H]
        gates = hb.gemm(x2, w_ih, trans_b=True, bias=bias).view(T, B, ndir, 4 * H)
        y = torch.empty(T, B, ndir * H, device=dev, dtype=torch.float32)
        c = torch.empty(T, B, ndir * H, device=dev, dtype=torch.float32)
        hb.lstm_seq_fwd(gates, w_hh, lens, y, c)
        ctx.save_for_backward(x2, w_ih, w_hh, gates, y, c, lens)
        ctx.dims = (T, B, I, H, ndir)
        ctx.mark_non_differentiable(lens)
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, w_ih, w_hh, gates, y, c, lens = ctx.saved_tensors
        T, B, I, H, ndir = ctx.dims
        dev = dy.device
        dy = dy.contiguous()
        w_hhT = w_hh.transpose(1, 2).contiguous()
        dcarry = torch.zeros(B, ndir * H, device=dev, dtype=torch.float32)
        hb.lstm_seq_bwd(gates, w_hhT, lens, dy, c, dcarry)        # gates now holds dG (in place)
        dG = gates.view(T * B, ndir * 4 * H)
        dx = hb.gemm(dG, w_ih).view(T, B, I) if ctx.needs_input_grad[0] else None
        dw_ih = hb.gemm(dG, x2, trans_a=True)                     # [ndir*4H, I]
        db = hb.colsum(dG)
        y2 = y.view(T * B, ndir * H)
        unperm = gate_unperm(H, dev)
        grads = []
        for d in range(ndir):
            if T > 1:
                if d == 0:     # h_{t-1} = y[t-1]
                    a = dG[B:, d * 4 * H:(d + 1) * 4 * H]
                    hprev = y2[:(T - 1) * B, d * H:(d + 1) * H]
                else:          # reverse direction: predecessor in processing order is y[t+1]
                    a = dG[:(T - 1) * B, d * 4 * H:(d + 1) * 4 * H]
                    hprev = y2[B:, d * H:(d + 1) * H]
                dw_hh = hb.gemm(a, hprev, trans_a=True)
            else:
                dw_hh = torch.zeros(4 * H, H, device=dev)
            dbd = db[d * 4 * H:(d + 1) * 4 * H][unperm]
            grads += [dw_ih[d * 4 * H:(d + 1) * 4 * H][unperm], dw_hh[unperm], dbd, dbd]
        return (dx, None, None) + tuple(grads)


def lstm_layer(x, lens, params, ndir):
    """x [T,B,I] time-major contiguous, lens int32 device [B] -> y [T,B,ndir*H]."""
    return _LstmLayer.apply(x.contiguous(), lens, ndir, *params)


# --------------------------------------------------------------------------------------
class _Pyramid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask):
        T, B, C = x.shape
        out = torch.empty((T + 1) // 2, B, 2 * C, device=x.device, dtype=torch.float32)
        hb.pyramid_fwd(x.contiguous(), mask, out)
        ctx.save_for_backward(mask)
        ctx.shape = (T, B, C)
        return out

    @staticmethod
    def backward(ctx, dout):
        (mask,) = ctx.saved_tensors
        din = torch.empty(ctx.shape, device=dout.device, dtype=torch.float32)
        hb.pyramid_bwd(dout.contiguous(), mask, din)
        return din, None


def pyramid_concat(x, mask=None):
    """[T,B,C] -> [ceil(T/2),B,2C] (model.py:85-92); `mask` = pre-scaled dropout mask or None."""
    return _Pyramid.apply(x, mask)


# --------------------------------------------------------------------------------------
def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class _DecoderSeq(torch.autograd.Function):
    """All decoder steps (Decoder.forward loop, model.py:324-351) as one graph node.

    inputs : P [B,Tp,A], Q [B,Tp,O] (= enc_h W_o^T, no bias), emb_w [V,E], w_ih [4D,E+O], w_hh [4D,D],
             b_ih, b_hh, wdec [A,D], convw [C,1,1,2K+1], watt [A,C], gvec [1,A], bo [O], w_out [V,D+O],
             b_out [V], w0 [B,Tp]
    opts   : dict(L, tokens [B,L] long or None, tf_flags list[bool] or None, smooth, smooth_scaling,
                  sample, scaling (attention temperature), xmask [L,B,O+E] or None)
    returns: logits [L,B,V], ws [L,B,Tp], prediction [L,B] (long)
    """

    @staticmethod
    def forward(ctx, P, Q, emb_w, w_ih, w_hh, b_ih, b_hh, wdec, convw, watt, gvec, bo, w_out, b_out, w0, opts):
        dev = P.device
        B, Tp, A = P.shape
        O = Q.shape[2]
        D = w_hh.shape[1]
        E = emb_w.shape[1]
        V = w_out.shape[0]
        C = convw.shape[0]
        K = (convw.shape[-1] - 1) // 2
        L = int(opts["L"])
        KX = D + O + E
        f32 = dict(device=dev, dtype=torch.float32)
        perm = gate_perm(D, dev)
        wcat = torch.cat([w_hh, w_ih[:, E:E + O], w_ih[:, :E]], 1)[perm].contiguous()      # [4D, KX]
        bcat = (b_ih + b_hh)[perm].contiguous()
        convw2 = convw.reshape(C, 2 * K + 1).contiguous()
        gv = gvec.reshape(A).contiguous()
        xmask = opts.get("xmask")
        P = P.contiguous()
        Q = Q.contiguous()
        w0 = w0.contiguous()
        wdec_c = wdec.contiguous()
        watt_c = watt.contiguous()
        bo_c = bo.contiguous()
        X = torch.zeros(L + 1, B, KX, **f32)
        Xd = torch.zeros(L + 1, B, KX, **f32) if xmask is not None else None     # dropout-masked operand copy
        buf = dict(
            gates=torch.empty(L, B, 4 * D, **f32), cstate=torch.empty(L, B, D, **f32),
            Dproj=torch.empty(L, B, A, **f32), fconv=torch.empty(L, B, C, Tp, **f32),
            S=torch.empty(L, B, Tp, A, **f32), energy=torch.empty(L, B, Tp, **f32),
            ws=torch.empty(L, B, Tp, **f32))
        fs = hb.DecFwd(B=B, Tp=Tp, A=A, D=D, O=O, E=E, C=C, K=K, L=L, scaling=float(opts.get("scaling", 2.0)),
                       P=_p(P), Q=_p(Q), bo=_p(bo_c), wcat=_p(wcat), bcat=_p(bcat), wdec=_p(wdec_c),
                       convw=_p(convw2), watt=_p(watt_c), gvec=_p(gv), w0=_p(w0), xmask=_p(xmask), X=_p(X), Xd=_p(Xd),
                       gates=_p(buf["gates"]), cstate=_p(buf["cstate"]), Dproj=_p(buf["Dproj"]),
                       fconv=_p(buf["fconv"]), S=_p(buf["S"]), energy=_p(buf["energy"]), ws=_p(buf["ws"]))
        lib = hb.load()
        tokens = opts.get("tokens")
        tf_flags = opts.get("tf_flags")
        smooth = bool(opts.get("smooth", False))
        sample = bool(opts.get("sample", False))
        all_teacher = tokens is not None and (tf_flags is None or all(tf_flags)) and not sample
        w_out_c = w_out.contiguous()
        fed = torch.zeros(L, B, dtype=torch.long, device=dev)       # token whose embedding fed step s (-1: smooth)
        probs_saved = []
        if all_teacher:
            fed.copy_(tokens.t())
            X[:L, :, D + O:] = emb_w[fed]
            if Xd is not None:
                Xd[:L, :, D + O:] = X[:L, :, D + O:] * xmask[:, :, O:]
            hb.check(lib.asr_dec_seq_fwd(ctypes.byref(fs), 0, L, hb.stream()), "asr_dec_seq_fwd")
            logits = hb.gemm(X[1:].view(L * B, KX)[:, :D + O], w_out_c, trans_b=True, bias=b_out).view(L, B, V)
            pred = logits.argmax(-1)
        else:
            logits = torch.empty(L, B, V, **f32)
            pred = torch.empty(L, B, dtype=torch.long, device=dev)
            for s in range(L):
                if s == 0:
                    tok = tokens[:, 0] if tokens is not None else torch.full((B,), opts["bos"], dtype=torch.long,
                                                                             device=dev)
                    fed[0] = tok
                    X[0, :, D + O:] = emb_w[tok]
                elif tokens is not None:
                    tok = tokens[:, s] if tf_flags[s] else pred[s - 1]
                    fed[s] = tok
                    X[s, :, D + O:] = emb_w[tok]
                elif not smooth:
                    fed[s] = pred[s - 1]
                    X[s, :, D + O:] = emb_w[pred[s - 1]]
                else:
                    pr = torch.softmax(logits[s - 1] * opts["smooth_scaling"], dim=-1)
                    probs_saved.append(pr)
                    fed[s] = -1
                    hb.gemm(pr, emb_w, out=X[s][:, D + O:])
                if Xd is not None:
                    Xd[s, :, D + O:] = X[s, :, D + O:] * xmask[s, :, O:]
                hb.check(lib.asr_dec_step_fwd(ctypes.byref(fs), s, hb.stream()), "asr_dec_step_fwd")
                hb.gemm_skinny(X[s + 1][:, :D + O], w_out_c, bias=b_out, out=logits[s])
                pred[s] = torch.distributions.Categorical(logits=logits[s]).sample() if sample \
                    else logits[s].argmax(-1)
        ctx.fs = fs
        ctx.keep = (P, Q, wcat, bcat, wdec_c, convw2, watt_c, gv, bo_c, w0, xmask, X, Xd, buf, fed, probs_saved,
                    w_out_c, emb_w)
        ctx.dims = (B, Tp, A, O, D, E, V, C, K, L, KX)
        ctx.smooth = smooth and tokens is None
        ctx.smooth_scaling = float(opts.get("smooth_scaling", 1.0))
        ctx.mark_non_differentiable(pred)
        return logits, buf["ws"], pred

    @staticmethod
    def backward(ctx, dlogits, dws, _dpred):
        (P, Q, wcat, bcat, wdec, convw2, watt, gv, bo, w0, xmask, X, Xd, buf, fed, probs_saved, w_out,
         emb_w) = ctx.keep
        B, Tp, A, O, D, E, V, C, K, L, KX = ctx.dims
        dev = P.device
        f32 = dict(device=dev, dtype=torch.float32)
        lib = hb.load()
        dlog2 = dlogits.contiguous().view(L * B, V)
        G = torch.zeros(L + 1, B, KX, **f32)
        XO = X[1:].view(L * B, KX)[:, :D + O]
        hb.gemm(dlog2, w_out, out=G[1:].view(L * B, KX)[:, :D + O])
        dw_out = hb.gemm(dlog2, XO, trans_a=True)
        db_out = hb.colsum(dlog2)
        ntile = (A + 63) // 64
        wk = dict(
            wcatT=wcat.t().contiguous(), wdecT=wdec.t().contiguous(), dwext=torch.zeros(C, B, Tp, **f32),
            dwraw=torch.empty(B, Tp, **f32), dfpart=torch.empty(ntile, B, C, Tp, **f32),
            dP=torch.zeros(B, Tp, A, **f32), dgates=torch.empty(L, B, 4 * D, **f32), dD=torch.empty(L, B, A, **f32),
            dcell=torch.zeros(B, D, **f32), dgvec_part=torch.zeros(B, A, **f32),
            dwatt_part=torch.zeros(B, A, C, **f32), dconv_part=torch.zeros(B, C, 2 * K + 1, **f32))
        dws_c = dws.contiguous() if dws is not None else None
        bs = hb.DecBwd(f=ctx.fs, wcatT=_p(wk["wcatT"]), wdecT=_p(wk["wdecT"]), dws=_p(dws_c), G=_p(G),
                       dwext=_p(wk["dwext"]), dwraw=_p(wk["dwraw"]), dfpart=_p(wk["dfpart"]), dP=_p(wk["dP"]),
                       dgates=_p(wk["dgates"]), dD=_p(wk["dD"]), dcell=_p(wk["dcell"]),
                       dgvec_part=_p(wk["dgvec_part"]), dwatt_part=_p(wk["dwatt_part"]),
                       dconv_part=_p(wk["dconv_part"]))
        demb_w = torch.zeros_like(emb_w)
        if not ctx.smooth:
            hb.check(lib.asr_dec_seq_bwd(ctypes.byref(bs), 0, L, hb.stream()), "asr_dec_seq_bwd")
        else:
            # smooth-embedding feedback (model.py:341): emb_s = softmax(logit_{s-1}*k) @ E couples step s to
            # the logits of step s-1, so the extra gradient is injected between the per-step kernels.
            k = ctx.smooth_scaling
            for s in range(L - 1, -1, -1):
                hb.check(lib.asr_dec_step_bwd(ctypes.byref(bs), s, hb.stream()), "asr_dec_step_bwd")
                if s >= 1:
                    demb = G[s][:, D + O:]
                    pr = probs_saved[s - 1]
                    hb.gemm(pr, demb, trans_a=True, out=demb_w, accumulate=True, split_k=1)
                    dp = hb.gemm(demb, emb_w, trans_b=True)
                    dl = (k * pr * (dp - (pr * dp).sum(-1, keepdim=True))).contiguous()
                    hb.gemm(dl, w_out, out=G[s][:, :D + O], accumulate=True, split_k=1)
                    hb.gemm(dl, X[s][:, :D + O], trans_a=True, out=dw_out, accumulate=True, split_k=1)
                    hb.colsum(dl, out=db_out, accumulate=True)
        # deferred weight gradients: one GEMM each over the whole sequence
        dg2 = wk["dgates"].view(L * B, 4 * D)
        Xin = X[:L] if Xd is None else Xd[:L]
        dwcat = hb.gemm(dg2, Xin.view(L * B, KX), trans_a=True)                  # [4D, KX] gate-interleaved rows
        unperm = gate_unperm(D, dev)
        dwcat = dwcat[unperm]
        dw_hh = dwcat[:, :D].contiguous()
        dw_ih = torch.cat([dwcat[:, D + O:], dwcat[:, D:D + O]], 1)
        dbias = hb.colsum(dg2)[unperm]
        dwdec = hb.gemm(wk["dD"].view(L * B, A), X[1:].view(L * B, KX)[:, :D], trans_a=True)
        dgvec = wk["dgvec_part"].sum(0).view(1, A)
        dwatt = wk["dwatt_part"].sum(0)
        dconvw = wk["dconv_part"].sum(0).view(C, 1, 1, 2 * K + 1)
        # dQ[b] = ws[:, b, :]^T dctx[:, b, :]   (batched over utterances)
        dQ = torch.empty(B, Tp, O, **f32)
        dctx_base = G[1:]                               # [L, B, KX], ctx grad at columns D:D+O
        hb.gemm_batched(buf["ws"], dctx_base[:, :, D:], dQ, True, False, Tp, O, L, B * Tp, B * KX, O, B, Tp, KX,
                        Tp * O)
        dbo = hb.colsum(dctx_base.view(L * B, KX)[:, D:D + O])
        # embedding gradient for token-fed steps
        demb_all = G[:L, :, D + O:]
        tokfed = fed >= 0
        if bool(tokfed.all()):
            demb_w.index_add_(0, fed.view(-1), demb_all.reshape(L * B, E))
        else:
            demb_w.index_add_(0, fed[tokfed], demb_all[tokfed])
        return (wk["dP"], dQ, demb_w, dw_ih, dw_hh, dbias, dbias, dwdec, dconvw, dwatt, dgvec, dbo, dw_out, db_out,
                None, None)


def decoder_sequence(P, Q, emb_w, w_ih, w_hh, b_ih, b_hh, wdec, convw, watt, gvec, bo, w_out, b_out, w0, opts):
    return _DecoderSeq.apply(P, Q, emb_w, w_ih, w_hh, b_ih, b_hh, wdec, convw, watt, gvec, bo, w_out, b_out, w0, opts)
