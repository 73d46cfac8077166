"""CPU ORACLE for the seq2seq-ASR training hot path.  TEST INFRASTRUCTURE ONLY.

This file is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The product path (``semi-supervised-asr_amd/``) never routes through
it and has no CPU fallback.

It restates, as plain explicit math over a flat ``{key: tensor}`` state dict
(keys = the reference ``state_dict`` keys), what the reference
computes with stock torch modules:

    pBLSTM / Encoder      /root/reference/model.py:58-112
    AttLoc                /root/reference/model.py:114-173
    Decoder               /root/reference/model.py:256-367
    E2E                   /root/reference/model.py:408-456
    LM (judge)            /root/reference/model.py:459-573
    train-step arithmetic /root/reference/solver.py:288-301, 375-385, 460-495
    host helpers          /root/reference/utils.py:134-235

Parity pin: every function here is checked by ``tests/test_oracle_golden.py``
against fixtures in ``tests/golden/*.npz`` that were produced by importing the
real reference in the build container (``tests/golden/make_golden.py``).

Everything is fp32 torch on CPU; autograd supplies the gradients the HIP
backward kernels are compared with.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

PAD, BOS, EOS = 0, 1, 2


# --------------------------------------------------------------------------
# host helpers (utils.py:173-190, 192-235)
# --------------------------------------------------------------------------
def pad_ragged(seqs, fill):
    """utils.py:173-179 (pad_list): ragged list of tensors -> [B, Lmax, ...]."""
    longest = max(int(s.shape[0]) for s in seqs)
    out = seqs[0].new_full((len(seqs), longest) + tuple(seqs[0].shape[1:]), fill)
    for row, s in enumerate(seqs):
        out[row, : s.shape[0]] = s
    return out


def length_mask(lengths, width):
    """utils.py:181-190 (_seq_mask): float 0/1 mask [B, width], 1 where col < len."""
    lens = torch.as_tensor(np.asarray(lengths), dtype=torch.long)
    return (torch.arange(width).unsqueeze(0) < lens.unsqueeze(1)).float()


def cut_at_eos(rows, eos=EOS):
    """utils.py:192-201 (remove_pad_eos): prefix of each row before first eos."""
    out = []
    for row in rows:
        row = list(row)
        out.append(row[: row.index(eos)] if eos in row else row)
    return out


def ids_to_sentences(rows, vocab, non_lang_syms):
    """utils.py:160-163,212-235 (to_sents): ids -> chars minus non-language
    symbols -> string, '<space>' rendered as ' '."""
    inv = {v: k for k, v in vocab.items()}
    drop = {vocab[s] for s in non_lang_syms}
    sents = []
    for row in rows:
        chars = [inv[int(i)] for i in row if int(i) not in drop]
        sents.append("".join(" " if ch == "<space>" else ch for ch in chars))
    return sents


def levenshtein(a, b):
    """Edit distance (what editdistance.eval returns at utils.py:225)."""
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def corpus_cer(hyps, refs):
    """utils.py:222-228 (calculate_cer): sum(edit) / sum(len(ref))."""
    dist = sum(levenshtein(h, r) for h, r in zip(hyps, refs))
    return dist / float(sum(len(r) for r in refs))


# --------------------------------------------------------------------------
# LSTM primitives (PyTorch gate order i,f,g,o)
# --------------------------------------------------------------------------
def lstm_direction(x, lens, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of torch.nn.LSTM run on a PackedSequence (model.py:79-81),
    restated as a masked recurrence on the padded tensor: state and output are
    0 at every (b, t) with t >= lens[b]."""
    bsz, steps, _ = x.shape
    hid = w_hh.shape[1]
    lens_t = torch.as_tensor(np.asarray(lens), dtype=torch.long)
    # unbind once instead of indexing per step: the backward of a per-step select allocates a zero tensor of the whole
    # [B,T,4H] sequence every step (the O(T^2) fill_/add_ that dominates the reference's own CPU step)
    gx = (x @ w_ih.t() + (b_ih + b_hh)).unbind(1)
    w_hh_t = w_hh.t()
    h = x.new_zeros(bsz, hid)
    c = x.new_zeros(bsz, hid)
    outs = [None] * steps
    order = range(steps - 1, -1, -1) if reverse else range(steps)
    for t in order:
        gates = gx[t] + h @ w_hh_t
        gi, gf, gg, go = gates.split(hid, dim=1)
        c_new = torch.sigmoid(gf) * c + torch.sigmoid(gi) * torch.tanh(gg)
        h_new = torch.sigmoid(go) * torch.tanh(c_new)
        live = (lens_t > t).to(x.dtype).unsqueeze(1)
        c = c_new * live
        h = h_new * live
        outs[t] = h
    return torch.stack(outs, dim=1)


def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.LSTMCell (model.py:262,286)."""
    hid = w_hh.shape[1]
    gates = x @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
    gi, gf, gg, go = gates.split(hid, dim=1)
    c_new = torch.sigmoid(gf) * c + torch.sigmoid(gi) * torch.tanh(gg)
    h_new = torch.sigmoid(go) * torch.tanh(c_new)
    return h_new, c_new


def pair_concat(y):
    """model.py:85-92: [B,T,C] -> [B,ceil(T/2),2C]; odd T replicates the last
    (padded) frame first."""
    if y.shape[1] % 2 == 1:
        y = torch.cat([y, y[:, -1:, :]], dim=1)
    return y.reshape(y.shape[0], y.shape[1] // 2, y.shape[2] * 2)


# --------------------------------------------------------------------------
# Encoder (model.py:58-112)
# --------------------------------------------------------------------------
def encoder_forward(sd, xs, ilens, n_layers, subsample, dropout_rate=0.0,
                    training=True, total_length=None, taps=None):
    """pBLSTM.forward.  `total_length` (not in the reference, which always pads
    to max(ilens)) lets a data-parallel shard keep the global padded length.
    `taps`, if a dict, receives per-layer intermediates for fixture checks."""
    x = xs
    lens = [int(l) for l in ilens]
    for i in range(n_layers):
        p = "encoder.enc2.layers.%d." % i
        steps = max(lens) if total_length is None else int(total_length[i])
        x = x[:, :steps]
        fwd = lstm_direction(x, lens, sd[p + "weight_ih_l0"], sd[p + "weight_hh_l0"],
                             sd[p + "bias_ih_l0"], sd[p + "bias_hh_l0"], False)
        bwd = lstm_direction(x, lens, sd[p + "weight_ih_l0_reverse"], sd[p + "weight_hh_l0_reverse"],
                             sd[p + "bias_ih_l0_reverse"], sd[p + "bias_hh_l0_reverse"], True)
        y = torch.cat([fwd, bwd], dim=2)
        if taps is not None:
            taps["lstm%d" % i] = y
        y = F.dropout(y, dropout_rate, training)
        sub = subsample[i]
        if sub > 1:
            y = pair_concat(y)
            lens = [(l + 1) // sub for l in lens]
            if taps is not None:
                taps["cat%d" % i] = y
        q = "encoder.enc2.project_layers.%d." % i
        x = torch.relu(y @ sd[q + "weight"].t() + sd[q + "bias"])
        if taps is not None:
            taps["proj%d" % i] = x
        x = F.dropout(x, dropout_rate, training)
    return x, lens


def padded_lengths(t_max, n_layers, subsample):
    """Padded time extent at the input of each encoder layer (and the output),
    for `total_length`: T_0 = t_max, T_{l+1} = ceil(T_l/2) when subsample>1."""
    out = [int(t_max)]
    for i in range(n_layers):
        out.append((out[-1] + 1) // 2 if subsample[i] > 1 else out[-1])
    return out


# --------------------------------------------------------------------------
# Location-aware attention (model.py:114-173)
# --------------------------------------------------------------------------
class AttState:
    """Per-batch cache of AttLoc (model.py:130-137,141-144)."""

    def __init__(self):
        self.enc_h = None
        self.pre = None


def attloc_step(sd, st, enc_pad, enc_len, dec_z, att_prev, scaling=2.0, prefix="attention."):
    """AttLoc.forward: returns (mlp_o(context) [B,odim], w [B,T'])."""
    bsz, frames, _ = enc_pad.shape
    if st.pre is None:
        st.enc_h = enc_pad
        st.pre = enc_pad @ sd[prefix + "mlp_enc.weight"].t() + sd[prefix + "mlp_enc.bias"]
    if att_prev is None:
        # model.py:151-153 uniform over the valid frames, 0 on the padding
        att_prev = enc_pad.new_zeros(bsz, frames)
        for b, l in enumerate(enc_len):
            att_prev[b, :l] = 1.0 / l
    filt = sd[prefix + "loc_conv.weight"]                      # [C,1,1,2K+1]
    half = (filt.shape[-1] - 1) // 2
    conv = F.conv1d(att_prev.unsqueeze(1), filt.reshape(filt.shape[0], 1, -1), padding=half)
    loc = conv.transpose(1, 2) @ sd[prefix + "mlp_att.weight"].t()        # [B,T',att]
    dec = (dec_z @ sd[prefix + "mlp_dec.weight"].t()).unsqueeze(1)        # [B,1,att]
    e = (torch.tanh(st.pre + dec + loc) @ sd[prefix + "gvec.weight"].t()).squeeze(2)
    w = torch.softmax(scaling * e, dim=1)          # over ALL T' frames
    ctx = torch.bmm(w.unsqueeze(1), st.enc_h).squeeze(1)
    out = ctx @ sd[prefix + "mlp_o.weight"].t() + sd[prefix + "mlp_o.bias"]
    return out, w


# --------------------------------------------------------------------------
# Decoder (model.py:256-367)
# --------------------------------------------------------------------------
def decoder_forward(sd, enc_pad, enc_len, ys=None, tf_rate=1.0, max_dec_timesteps=500,
                    sample=False, smooth=False, scaling=1.0, label_smoothing=True,
                    training=True, ls_weight=0.0, labeldist=None, dropout_rate=0.0,
                    olength_override=None):
    """Decoder.forward -> (logits, ys_log_probs, prediction, ws).
    `olength_override` (not in the reference) lets a data-parallel shard decode
    the global number of steps: targets are EOS-padded to it."""
    bsz = enc_pad.shape[0]
    emb_w = sd["decoder.embedding.weight"]
    dec_dim = sd["decoder.LSTMCell.weight_hh"].shape[1]
    odim = sd["attention.mlp_o.weight"].shape[0]
    have_ys = ys is not None and len(ys) > 0
    if ys is not None:
        bos = ys[0].new_tensor([BOS])
        eos = ys[0].new_tensor([EOS])
        tgt_in = pad_ragged([torch.cat([bos, y]) for y in ys], EOS)
        tgt_out = pad_ragged([torch.cat([y, eos]) for y in ys], EOS)
        if olength_override is not None and olength_override > tgt_out.shape[1]:
            extra = olength_override - tgt_out.shape[1]
            tgt_in = F.pad(tgt_in, (0, extra), value=EOS)
            tgt_out = F.pad(tgt_out, (0, extra), value=EOS)
        olength = tgt_out.shape[1]
        eys = emb_w[tgt_in].unbind(1)
    z = enc_pad.new_zeros(bsz, dec_dim)
    cstate = enc_pad.new_zeros(bsz, dec_dim)
    ctx = enc_pad.new_zeros(bsz, odim)
    w = None
    st = AttState()
    logits, preds, ws = [], [], []
    if not have_ys:
        olength = max_dec_timesteps
    logit = None
    for t in range(olength):
        if ys is not None:
            # one numpy draw per step even at tf_rate=1
            use_truth = np.random.random_sample() <= tf_rate
            emb = eys[t] if (use_truth or t == 0) else emb_w[preds[-1]]
        elif t == 0:
            emb = emb_w[torch.full((bsz,), BOS, dtype=torch.long)]
        elif not smooth:
            emb = emb_w[preds[-1]]
        else:
            emb = torch.softmax(logit * scaling, dim=-1) @ emb_w
        cell_in = F.dropout(torch.cat([emb, ctx], dim=-1), dropout_rate, training)
        z, cstate = lstm_cell(cell_in, z, cstate,
                              sd["decoder.LSTMCell.weight_ih"], sd["decoder.LSTMCell.weight_hh"],
                              sd["decoder.LSTMCell.bias_ih"], sd["decoder.LSTMCell.bias_hh"])
        ctx, w = attloc_step(sd, st, enc_pad, enc_len, z, w)     # default scaling 2.0
        logit = torch.cat([z, ctx], dim=-1) @ sd["decoder.output_layer.weight"].t() \
            + sd["decoder.output_layer.bias"]
        ws.append(w)
        logits.append(logit)
        if sample:
            preds.append(torch.distributions.Categorical(logits=logit).sample())
        else:
            preds.append(torch.argmax(logit, dim=-1))
    logits = torch.stack(logits, dim=1)
    log_probs = torch.log_softmax(logits, dim=2)
    prediction = torch.stack(preds, dim=1)
    ws = torch.stack(ws, dim=1)
    index = tgt_out if have_ys else prediction
    ys_lp = torch.gather(log_probs, 2, index.unsqueeze(2)).squeeze(2)
    if label_smoothing and ls_weight > 0 and training:
        dist = torch.as_tensor(np.asarray(labeldist, dtype=np.float32))
        ys_lp = (1 - ls_weight) * ys_lp + ls_weight * torch.sum(log_probs * dist, dim=2)
    return logits, ys_lp, prediction, ws


# --------------------------------------------------------------------------
# E2E (model.py:408-456)
# --------------------------------------------------------------------------
def e2e_forward(sd, cfg, xs, ilens, ys=None, tf_rate=1.0, max_dec_timesteps=200, sample=False,
                smooth=False, scaling=1.0, label_smoothing=True, training=True,
                total_length=None, olength_override=None):
    """E2E.forward.  cfg keys: enc_n_layers, subsample, dropout_rate, ls_weight, labeldist."""
    enc_h, enc_lens = encoder_forward(sd, xs, ilens, cfg["enc_n_layers"], cfg["subsample"],
                                      cfg.get("dropout_rate", 0.0), training, total_length)
    return decoder_forward(sd, enc_h, enc_lens, ys, tf_rate=tf_rate,
                           max_dec_timesteps=max_dec_timesteps, sample=sample, smooth=smooth,
                           scaling=scaling, label_smoothing=label_smoothing, training=training,
                           ls_weight=cfg.get("ls_weight", 0.0), labeldist=cfg.get("labeldist"),
                           dropout_rate=cfg.get("dropout_rate", 0.0),
                           olength_override=olength_override)


def masked_loss(log_probs, ys):
    """E2E.mask_and_cal_loss with mask=None (model.py:447-456): lengths +1 for EOS."""
    lens = [int(y.shape[0]) + 1 for y in ys]
    return -torch.sum(log_probs * length_mask(lens, log_probs.shape[1])) / sum(lens)


# --------------------------------------------------------------------------
# LM judge (model.py:459-573)
# --------------------------------------------------------------------------
def lm_forward(sd, ys, discrete_input=True, n_layers=2, dropout_rate=0.0, training=True,
               ls_weight=0.0, labeldist=None):
    """LM.forward -> (ys_log_probs, ys_probs, predictions)."""
    emb_w = sd["embedding.weight"]
    if discrete_input:
        bos = ys[0].new_tensor([BOS])
        eos = ys[0].new_tensor([EOS])
        seq_in = [torch.cat([bos, y, eos, eos, eos, eos]) for y in ys]
        seq_out = [torch.cat([y, eos, eos, eos, eos, eos]) for y in ys]
        tok_in = pad_ragged(seq_in, EOS)
        tok_out = pad_ragged(seq_out, EOS)
        lens = [int(s.shape[0]) for s in seq_in]
    else:
        first = torch.full((ys.shape[0], 1), BOS, dtype=ys.dtype)
        tok_in = torch.cat([first, ys[:, :-1]], dim=1)
        tok_out = ys
        lens = [tok_in.shape[1]] * tok_in.shape[0]
    x = F.dropout(emb_w[tok_in], dropout_rate, training)
    for l in range(n_layers):
        x = lstm_direction(x, lens, sd["LSTM.weight_ih_l%d" % l], sd["LSTM.weight_hh_l%d" % l],
                           sd["LSTM.bias_ih_l%d" % l], sd["LSTM.bias_hh_l%d" % l], False)
        if l + 1 < n_layers:
            x = F.dropout(x, dropout_rate, training)      # nn.LSTM inter-layer dropout
    x = F.dropout(x, dropout_rate, training)
    logits = x @ sd["output_layer.weight"].t() + sd["output_layer.bias"]
    log_probs = torch.log_softmax(logits, dim=2)
    probs = torch.softmax(logits, dim=2)
    ys_lp = torch.gather(log_probs, 2, tok_out.unsqueeze(2)).squeeze(2)
    ys_p = torch.gather(probs, 2, tok_out.unsqueeze(2)).squeeze(2)
    if ls_weight > 0 and training:
        dist = torch.as_tensor(np.asarray(labeldist, dtype=np.float32))
        ys_lp = (1 - ls_weight) * ys_lp + ls_weight * torch.sum(log_probs * dist, dim=2)
    return ys_lp, ys_p, torch.argmax(logits, dim=-1)


def lm_masked_sum(values, ys):
    """LM.mask_and_cal_sum with mask=None (model.py:565-573): lengths +1+4."""
    lens = [int(y.shape[0]) + 5 for y in ys]
    return torch.sum(values * length_mask(lens, values.shape[1])) / sum(lens)


# --------------------------------------------------------------------------
# Optimiser arithmetic (torch.optim.Adam(amsgrad=True, weight_decay) +
# clip_grad_norm_, as called at solver.py:152-153,384-385)
# --------------------------------------------------------------------------
def clip_global_norm(grads, max_norm):
    """clip_grad_norm_: scale all grads by min(1, max_norm/(||g||_2 + 1e-6))."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return [g * coef for g in grads], total


class AdamAmsgrad:
    """Adam with L2 weight decay folded into the gradient and AMSGrad's running
    max of the second moment (defaults betas=(0.9,0.999), eps=1e-8)."""

    def __init__(self, names, lr, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8):
        self.names = list(names)
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.t = 0
        self.m, self.v, self.vmax = {}, {}, {}

    def step(self, sd, grads):
        b1, b2 = self.betas
        self.t += 1
        c1 = 1 - b1 ** self.t
        c2 = 1 - b2 ** self.t
        with torch.no_grad():
            for name in self.names:
                p, g = sd[name], grads[name]
                if self.wd:
                    g = g + self.wd * p
                if name not in self.m:
                    self.m[name] = torch.zeros_like(p)
                    self.v[name] = torch.zeros_like(p)
                    self.vmax[name] = torch.zeros_like(p)
                self.m[name].mul_(b1).add_(g, alpha=1 - b1)
                self.v[name].mul_(b2).addcmul_(g, g, value=1 - b2)
                torch.maximum(self.vmax[name], self.v[name], out=self.vmax[name])
                denom = self.vmax[name].sqrt() / math.sqrt(c2) + self.eps
                p.addcdiv_(self.m[name], denom, value=-self.lr / c1)


def unique_param_names(sd):
    """The attention weights appear twice in the state dict; the
    optimiser sees each tensor once (first name wins, like named_parameters)."""
    seen, names = set(), []
    for k, v in sd.items():
        if id(v) not in seen:
            seen.add(id(v))
            names.append(k)
    return names


def make_leaf_state(arrays):
    """numpy dict -> torch leaf tensors (requires_grad), sharing one tensor for
    the duplicated `decoder.attention.*` / `attention.*` keys."""
    sd = {}
    for k, a in arrays.items():
        if k.startswith("decoder.attention."):
            continue
        sd[k] = torch.tensor(np.asarray(a), dtype=torch.float32, requires_grad=True)
    for k in list(sd):
        if k.startswith("attention."):
            sd["decoder." + k] = sd[k]
    return sd


def sup_train_step(sd, cfg, opt, xs, ilens, ys, tf_rate=1.0, max_grad_norm=5.0,
                   loss_scale=None, total_length=None, olength_override=None):
    """One iteration of Solver.sup_train_one_epoch (solver.py:375-385):
    loss = -mean(log_probs) over the whole [B, olength] grid."""
    names = unique_param_names(sd)
    _, lp, _, _ = e2e_forward(sd, cfg, xs, ilens, ys, tf_rate=tf_rate, training=True,
                              total_length=total_length, olength_override=olength_override)
    loss = -torch.mean(lp) if loss_scale is None else -torch.sum(lp) * loss_scale
    grads = torch.autograd.grad(loss, [sd[n] for n in names])
    clipped, gnorm = clip_global_norm(list(grads), max_grad_norm)
    opt.step(sd, dict(zip(names, clipped)))
    return float(loss.detach()), float(gnorm), dict(zip(names, grads))


def ssl_losses(sd, jsd, cfg, jcfg, lab_xs, lab_ilens, lab_ys, unlab_xs, unlab_ilens,
               proportion, smooth=True, scaling=3.0):
    """Loss assembly of Solver.gen_train_one_iteration (solver.py:465-483)."""
    _, u_lp, u_pred, _ = e2e_forward(
        sd, cfg, unlab_xs, unlab_ilens, ys=None, sample=False, label_smoothing=False,
        max_dec_timesteps=int(unlab_xs.shape[1] * proportion), smooth=smooth, scaling=scaling)
    _, lm_p, _ = lm_forward(jsd, u_pred, discrete_input=False, n_layers=jcfg["n_layers"],
                            dropout_rate=jcfg.get("dropout_rate", 0.0), training=True,
                            ls_weight=jcfg.get("ls_weight", 0.0), labeldist=jcfg.get("labeldist"))
    mask = (u_pred != EOS).float()
    unsup = -torch.sum(lm_p * u_lp * mask) / torch.sum(mask)
    _, l_lp, _, _ = e2e_forward(sd, cfg, lab_xs, lab_ilens, lab_ys, tf_rate=1.0)
    sup = -torch.mean(l_lp)
    return sup, unsup
