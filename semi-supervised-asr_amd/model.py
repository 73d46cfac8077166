"""MI355X-native seq2seq ASR model behind the reference's class surface.

Same class names, constructor signatures, forward signatures/returns and state_dict keys as
/root/reference/model.py (pBLSTM 58-98, Encoder 100-112, AttLoc 114-173, Decoder 256-367,
E2E 408-456, LM 459-573) — but every hot operator is a hand-written gfx950 kernel reached
through ops.py / hip_backend.py.  Activations are time-major inside the encoder, the decoder
loop is one fused graph node, and mlp_o is hoisted out of the step loop.
There is no CPU execution path: tensors must be on the GPU.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

import hip_backend as hb
import ops
from hip_backend import to_device_i32 as hb_to_device
from utils import cc, pad_list, _seq_mask


class _LstmWeights(torch.nn.Module):
    """Parameter container with torch.nn.LSTM's names / shapes / default init (U(-1/sqrt(H), 1/sqrt(H)))
    so checkpoints interchange with the reference.  It has no forward of its own."""

    def __init__(self, input_dim, hidden_dim, num_layers=1, bidirectional=False):
        super().__init__()
        self.input_size, self.hidden_size = input_dim, hidden_dim
        self.num_layers, self.bidirectional = num_layers, bidirectional
        k = 1.0 / math.sqrt(hidden_dim)
        for layer in range(num_layers):
            idim = input_dim if layer == 0 else hidden_dim * (2 if bidirectional else 1)
            for suffix in ([""] + (["_reverse"] if bidirectional else [])):
                for name, shape in (("weight_ih", (4 * hidden_dim, idim)), ("weight_hh", (4 * hidden_dim, hidden_dim)),
                                    ("bias_ih", (4 * hidden_dim,)), ("bias_hh", (4 * hidden_dim,))):
                    p = torch.nn.Parameter(torch.empty(*shape).uniform_(-k, k))
                    self.register_parameter("%s_l%d%s" % (name, layer, suffix), p)

    def direction_params(self, layer):
        out = []
        for suffix in ([""] + (["_reverse"] if self.bidirectional else [])):
            out += [getattr(self, "%s_l%d%s" % (n, layer, suffix)) for n in
                    ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        return out


class _CellWeights(torch.nn.Module):
    """torch.nn.LSTMCell-shaped parameter container (weight_ih, weight_hh, bias_ih, bias_hh)."""

    def __init__(self, input_dim, hidden_dim):
        super().__init__()
        k = 1.0 / math.sqrt(hidden_dim)
        self.weight_ih = torch.nn.Parameter(torch.empty(4 * hidden_dim, input_dim).uniform_(-k, k))
        self.weight_hh = torch.nn.Parameter(torch.empty(4 * hidden_dim, hidden_dim).uniform_(-k, k))
        self.bias_ih = torch.nn.Parameter(torch.empty(4 * hidden_dim).uniform_(-k, k))
        self.bias_hh = torch.nn.Parameter(torch.empty(4 * hidden_dim).uniform_(-k, k))


def _drop_mask(shape, p, device):
    """Inverted-dropout mask, already scaled by 1/(1-p): on the GPU a (seed, p) descriptor that the consuming kernels
    expand in flight (hip_backend.SeededMask), otherwise / with ASR_SEEDED_DROPOUT=0 a materialised tensor.  Tests
    replace this function to inject given masks."""
    if hb.USE_SEEDED_DROPOUT and torch.device(device).type == "cuda":
        return hb.SeededMask(shape, p, device)
    return torch.empty(shape, device=device, dtype=torch.float32).bernoulli_(1.0 - p).mul_(1.0 / (1.0 - p))


def _mask_tensor(mask):
    return mask.tensor() if isinstance(mask, hb.SeededMask) else mask


def padded_lengths(t_max, n_layers, subsample):
    """Padded time extent entering each encoder layer (+ the output extent)."""
    out = [int(t_max)]
    for i in range(n_layers):
        out.append((out[-1] + 1) // 2 if subsample[i] > 1 else out[-1])
    return out


class pBLSTM(torch.nn.Module):
    """Pyramidal BiLSTM stack (model.py:58-98)."""

    def __init__(self, input_dim, hidden_dim, n_layers, subsample, dropout_rate):
        super(pBLSTM, self).__init__()
        layers, project_layers = [], []
        for i in range(n_layers):
            idim = input_dim if i == 0 else hidden_dim
            project_dim = hidden_dim * 4 if subsample[i] > 1 else hidden_dim * 2
            layers.append(_LstmWeights(idim, hidden_dim, num_layers=1, bidirectional=True))
            project_layers.append(torch.nn.Linear(project_dim, hidden_dim))
        self.layers = torch.nn.ModuleList(layers)
        self.project_layers = torch.nn.ModuleList(project_layers)
        self.dropout_layer = torch.nn.Dropout(p=dropout_rate)
        self.subsample = subsample
        self.dropout_rate = dropout_rate

    def forward(self, xpad, ilens, total_length=None):
        """xpad [B,T,idim] zero padded, ilens descending host ints -> ([B,T',H], list[int]).
        `total_length` (list from padded_lengths) keeps a data-parallel shard at the global padded
        extent; default = max(ilens) like pad_packed_sequence (model.py:81).

        Default: PACKED ROWS (hb.RowLayout; include/asr_hip.h) - what pack_padded_sequence gives the reference's LSTM
        (model.py:79-81), kept for the whole stack: every utterance is a block of consecutive rows (its frames + at least one
        zero row), every product of the encoder a GEMM over sum(len) rows instead of B * T_max, the pair-concat a reshape.
        The padded batch only exists at the two ends: the collated input is packed by one kernel, the output is unpacked into
        the [B, T', H] tensor the decoder reads, its frames behind an utterance filled with what the reference computes
        there, dropout(relu(bias)) of the last projection.  ASR_ENCODER_ROWS=padded: the time-major padded
        path (measurement)."""
        if hb.USE_PACKED_ROWS and xpad.is_cuda and xpad.shape[2] % 4 == 0:
            return self._forward_packed(xpad, ilens, total_length)
        return self._forward_padded(xpad, ilens, total_length)

    def _forward_packed(self, xpad, ilens, total_length):
        dev = xpad.device
        layout = hb.RowLayout([int(l) for l in ilens], [self.subsample[i] for i in range(len(self.layers))], dev,
                              t_pad=total_length)
        drop = self.training and self.dropout_rate > 0
        p = self.dropout_rate
        x = hb.rows_pack(xpad if xpad.is_contiguous() else xpad.contiguous(), hb.LayerRows(layout, 0))      # [R_0, idim]
        packs = ops.lstm_pack([layer.direction_params(0) for layer in self.layers], 2)   # kernel layout, all layers: one launch
        for i, (layer, proj) in enumerate(zip(self.layers, self.project_layers)):
            rows = hb.LayerRows(layout, i)
            y = ops.lstm_layer(x, None, None, 2, rows=rows, packed=packs[i])                     # [R_i, 2H]
            mask = _drop_mask((rows.R, 1, y.shape[1]), p, dev) if drop else None
            if self.subsample[i] > 1:
                rep = layout.replicated_rows(i)
                rep = hb.to_device_i64(rep, dev) if rep else None
                y = ops.pyramid_concat(y.view(rows.R, 1, -1), mask, rep).view(rows.R // 2, -1)  # [R_{i+1}, 4H]
            elif mask is not None:
                y = y * _mask_tensor(mask).view_as(y)
            if drop:
                m2 = _drop_mask((y.shape[0], 1, proj.weight.shape[0]), p, dev)
                if isinstance(m2, hb.SeededMask) and y.shape[0] * proj.weight.shape[0] % 4 == 0:
                    x = ops.linear(y, proj.weight, proj.bias, relu=True, drop=m2)      # relu -> dropout in the op
                else:
                    x = ops.linear(y, proj.weight, proj.bias, relu=True) * _mask_tensor(m2).view(y.shape[0], -1)
            else:
                x = ops.linear(y, proj.weight, proj.bias, relu=True)
        n = len(self.layers)
        t_out = layout.t_pad[n]
        pad_mask = _drop_mask((layout.B, t_out, x.shape[1]), p, dev) if drop else None
        # (frames behind an utterance: dropout(relu(bias)) of the last projection, SURVEY F2 - the relu is the kernel's)
        out = ops.rows_unpack(x, hb.LayerRows(layout, n), t_out, self.project_layers[-1].bias, pad_mask, fill_relu=True)
        self.last_lens_dev = layout.lens_dev(n)                    # device copy of the output lengths
        self.last_layout = layout                                  # (tests: where each utterance's rows were)
        return out, [int(l) for l in layout.lens[n]]

    def _forward_padded(self, xpad, ilens, total_length=None):
        dev = xpad.device
        lens = [int(l) for l in ilens]
        # lengths entering every layer are known on the host up front: one non-blocking upload for all layers
        per_layer = [lens]
        for i in range(len(self.layers)):
            sub = self.subsample[i]
            per_layer.append([(l + 1) // sub for l in per_layer[-1]] if sub > 1 else per_layer[-1])
        lens_all = hb_to_device(per_layer, dev)                    # [n_layers+1, B] int32
        x = xpad.transpose(0, 1)                                   # time-major
        drop = self.training and self.dropout_rate > 0
        packs = ops.lstm_pack([layer.direction_params(0) for layer in self.layers], 2)
        for i, (layer, proj) in enumerate(zip(self.layers, self.project_layers)):
            steps = max(lens) if total_length is None else int(total_length[i])
            if steps != x.shape[0]:                                 # (a no-op slice still records a SliceBackward:
                x = x[:steps]                                      #  a zero fill + a copy per layer in the backward)
            if not x.is_contiguous():
                x = x.contiguous()
            lens_dev = lens_all[i]
            y = ops.lstm_layer(x, lens_dev, None, 2, packed=packs[i])          # [T,B,2H]
            mask = _drop_mask(y.shape, self.dropout_rate, dev) if drop else None
            sub = self.subsample[i]
            if sub > 1:
                y = ops.pyramid_concat(y, mask)                    # [ceil(T/2),B,4H]
                lens = [(l + 1) // sub for l in lens]
            elif mask is not None:
                y = y * _mask_tensor(mask)
            if drop:
                m2 = _drop_mask((y.shape[0], y.shape[1], proj.weight.shape[0]), self.dropout_rate, dev)
                if isinstance(m2, hb.SeededMask) and m2.shape[0] * m2.shape[1] * m2.shape[2] % 4 == 0:
                    x = ops.linear(y, proj.weight, proj.bias, relu=True, drop=m2)      # relu -> dropout in the op
                else:
                    x = ops.linear(y, proj.weight, proj.bias, relu=True) * _mask_tensor(m2)
            else:
                x = ops.linear(y, proj.weight, proj.bias, relu=True)
        self.last_lens_dev = lens_all[-1]                          # device copy of the output lengths
        return x.transpose(0, 1).contiguous(), [int(l) for l in lens]


class Encoder(torch.nn.Module):
    """model.py:100-112 (pass-through to enc2; the VGG front end is dead code in the reference)."""

    def __init__(self, input_dim, hidden_dim, n_layers, subsample, dropout_rate, in_channel=1):
        super(Encoder, self).__init__()
        self.enc2 = pBLSTM(input_dim=input_dim, hidden_dim=hidden_dim, n_layers=n_layers, subsample=subsample,
                           dropout_rate=dropout_rate)

    def forward(self, x, ilens, total_length=None):
        return self.enc2(x, ilens, total_length)


class AttLoc(torch.nn.Module):
    """Location-aware attention parameters (model.py:114-137).  The per-step arithmetic of
    AttLoc.forward (139-173) runs inside the fused decoder sequence (ops.decoder_sequence); this
    module owns the weights under the reference's names."""

    def __init__(self, encoder_dim, decoder_dim, att_dim, conv_channels, conv_kernel_size, att_odim):
        super(AttLoc, self).__init__()
        self.mlp_enc = torch.nn.Linear(encoder_dim, att_dim)
        self.mlp_dec = torch.nn.Linear(decoder_dim, att_dim, bias=False)
        self.mlp_att = torch.nn.Linear(conv_channels, att_dim, bias=False)
        self.loc_conv = torch.nn.Conv2d(1, conv_channels, (1, 2 * conv_kernel_size + 1),
                                        padding=(0, conv_kernel_size), bias=False)
        self.gvec = torch.nn.Linear(att_dim, 1, bias=False)
        self.mlp_o = torch.nn.Linear(encoder_dim, att_odim)
        self.encoder_dim, self.decoder_dim = encoder_dim, decoder_dim
        self.att_dim, self.att_odim, self.conv_channels = att_dim, att_odim, conv_channels
        self.reset()

    def reset(self):
        self.enc_length = None
        self.enc_h = None
        self.pre_compute_enc_h = None

    @staticmethod
    def initial_weights(enc_len, frames, device):
        """model.py:151-153: uniform over each utterance's valid frames, 0 on the padding (no host stall in the
        middle of the step: pinned staging + non-blocking copy)."""
        if torch.device(device).type != "cuda":
            lens = torch.tensor([float(l) for l in enc_len]).unsqueeze(1)
            grid = torch.arange(frames, dtype=torch.float32).unsqueeze(0)
            return (grid < lens).to(torch.float32) / lens
        w0 = np.zeros((len(enc_len), int(frames)), dtype=np.float32)     # the lengths are host ints: build it there,
        for b, l in enumerate(enc_len):                                 # one non-blocking upload instead of 5 launches
            w0[b, :int(l)] = np.float32(1.0) / np.float32(int(l))
        return hb.to_device_f32(w0, device)

    def forward(self, enc_pad, enc_len, dec_z, att_prev, scaling=2.0):
        """Single attention step with the reference's signature (model.py:139-173): returns (mlp_o(context), w).
        Caches mlp_enc(enc_h) and enc_h W_o^T until reset(), like the reference caches pre_compute_enc_h.
        Forward only: training differentiates through the fused decoder sequence instead."""
        bsz, frames, _ = enc_pad.shape
        if self.pre_compute_enc_h is None:
            self.enc_h = enc_pad
            self.enc_length = frames
            with torch.no_grad():
                self.pre_compute_enc_h = ops.linear(enc_pad, self.mlp_enc.weight, self.mlp_enc.bias)
                self._q = ops.linear(enc_pad, self.mlp_o.weight, None)
        if dec_z is None:
            dec_z = enc_pad.new_zeros(bsz, self.decoder_dim)
        if att_prev is None:
            att_prev = AttLoc.initial_weights(enc_len, frames, enc_pad.device)
        return ops.attention_step(enc_pad, self.pre_compute_enc_h, self._q, self.mlp_dec.weight, self.loc_conv.weight,
                                  self.mlp_att.weight, self.gvec.weight, self.mlp_o.bias,
                                  dec_z.view(bsz, self.decoder_dim), att_prev, scaling)


class Decoder(torch.nn.Module):
    """model.py:256-367."""

    def __init__(self, output_dim, embedding_dim, hidden_dim, attention, att_odim, dropout_rate, bos, eos, pad,
                 ls_weight=0, labeldist=None):
        super(Decoder, self).__init__()
        self.bos, self.eos, self.pad = bos, eos, pad
        self.embedding = torch.nn.Embedding(output_dim, embedding_dim, padding_idx=pad)
        self.LSTMCell = _CellWeights(embedding_dim + att_odim, hidden_dim)
        self.output_layer = torch.nn.Linear(hidden_dim + att_odim, output_dim)
        self.dropout_layer = torch.nn.Dropout(p=dropout_rate)
        self.attention = attention
        self.hidden_dim, self.att_odim, self.dropout_rate = hidden_dim, att_odim, dropout_rate
        self._tok_const = {}
        self._dist_dev = {}
        self.ls_weight = ls_weight
        self.labeldist = labeldist
        if labeldist is not None:
            # plain attribute, not a buffer -> absent from state_dict
            self.vlabeldist = cc(torch.from_numpy(np.array(labeldist, dtype=np.float32)))

    def zero_state(self, enc_pad, dim=None):
        """model.py:276-280."""
        return enc_pad.new_zeros(enc_pad.size(0), dim if dim else self.hidden_dim)

    def forward_step(self, emb, dec_z, dec_c, c, w, enc_pad, enc_len):
        """One decoder step with the reference's signature (model.py:283-294): dropout(cat[emb, c]) -> LSTMCell ->
        attention -> output layer; returns (logit, dec_z, dec_c, c, w).  Thin, forward-only entry for callers that drive
        the loop themselves (inspection, custom search): the products run on the library's GEMM and attention-step
        kernels, the gate arithmetic on torch elementwise ops.  Training and Decoder.forward never come through here -
        they run all steps inside the fused sequence kernels (ops.decoder_sequence), which is what the fixtures pin; the
        tests hold this method to that path."""
        with torch.no_grad():
            cell = self.LSTMCell
            cell_inp = self.dropout_layer(torch.cat([emb, c], dim=-1)).contiguous()
            gates = hb.gemm(cell_inp, cell.weight_ih, trans_b=True, bias=cell.bias_ih)
            hb.gemm(dec_z.contiguous(), cell.weight_hh, trans_b=True, bias=cell.bias_hh, out=gates, accumulate=True)
            gi, gf, gg, go = gates.chunk(4, dim=1)
            dec_c = torch.sigmoid(gf) * dec_c + torch.sigmoid(gi) * torch.tanh(gg)
            dec_z = torch.sigmoid(go) * torch.tanh(dec_c)
            c, w = self.attention(enc_pad, enc_len, dec_z, w)
            logit = hb.gemm(torch.cat([dec_z, c], dim=-1).contiguous(), self.output_layer.weight, trans_b=True,
                            bias=self.output_layer.bias)
        return logit, dec_z, dec_c, c, w

    def _label_matrices(self, ys, olength=None):
        """ys_in = [BOS, y], ys_out = [y, EOS], both padded with EOS (model.py:301-306): ys_in as a [B, L] matrix, ys_out
        as [B, L] (a view) AND in the time-major [L, B] order the loss kernel indexes the logits with (third result).
        One concatenation + one gather on the device with indices built on the host from the (host-known) label
        lengths, instead of 2B concatenations and 2B row copies that leave the GPU idle behind the launch queue."""
        lens = [int(y.size(0)) for y in ys]
        bsz, n = len(ys), int(sum(lens))
        steps = max(lens) + 1
        if olength is not None and olength > steps:
            steps = int(olength)
        dev = ys[0].device
        key = (str(dev), str(ys[0].dtype))
        const = self._tok_const.get(key)
        if const is None:
            const = torch.tensor([self.bos, self.eos], dtype=ys[0].dtype, device=dev)
            self._tok_const[key] = const
        flat = torch.cat([y.reshape(-1) for y in ys] + [const])          # [n + 2]; n = BOS slot, n + 1 = EOS slot
        idx = np.full((2, bsz * steps), n + 1, dtype=np.int32)
        t_in, t_out = idx[0].reshape(bsz, steps), idx[1].reshape(steps, bsz)       # [B, L] and [L, B]
        off = 0
        for b, ln in enumerate(lens):
            t_in[b, 0] = n
            t_in[b, 1:1 + ln] = np.arange(off, off + ln)
            t_out[:ln, b] = np.arange(off, off + ln)
            off += ln
        if dev.type == "cuda":
            didx = hb.to_device_i64(idx, dev)
        else:
            didx = torch.from_numpy(idx).to(torch.long)
        both = flat[didx]
        out_lb = both[1].view(steps, bsz)
        return both[0].view(bsz, steps), out_lb.t(), out_lb

    def forward(self, enc_pad, enc_len, ys=None, tf_rate=1.0, max_dec_timesteps=500, sample=False, smooth=False,
                scaling=1.0, label_smoothing=True, olength=None, loss_norm=None):
        """-> (logits [B,L,V], ys_log_probs [B,L], prediction [B,L], ws [B,L,T']).
        `olength` (not in the reference) forces the number of teacher-forced steps so every
        data-parallel shard decodes the global olength.
        `loss_norm` (not in the reference): the number of utterances B the caller's loss -sum(ys_log_probs) / (B L)
        (solver.py:377) divides by - the kernel that forms ys_log_probs then leaves that loss in `ys_log_probs.fused_loss`
        (a device scalar with the graph behind it; parallel.local_loss returns it)."""
        dev = enc_pad.device
        bsz, frames, _ = enc_pad.shape
        att = self.attention
        att.reset()
        have_ys = ys is not None and len(ys) > 0
        opts = dict(scaling=2.0, smooth=bool(smooth), smooth_scaling=float(scaling), sample=bool(sample),
                    bos=self.bos, eos=self.eos)      # attention temperature is the AttLoc default
        if ys is not None:
            tok_in, tok_out, tok_out_lb = self._label_matrices(ys, olength)
            steps = tok_out.size(1)
            # one numpy draw per step, also at tf_rate=1 and for step 0 (model.py:328)
            draws = [np.random.random_sample() <= tf_rate for _ in range(steps)]
            draws[0] = True
            # every step teacher-forced: the argmax of the logits is only an output, and the loss kernel below reads them anyway
            opts.update(tokens=tok_in.to(dev), tf_flags=draws, skip_pred=have_ys and all(draws) and not sample)
        if not have_ys:
            steps = max_dec_timesteps
        opts["L"] = steps
        p = self.dropout_rate
        if self.training and p > 0:
            opts["xmask"] = _mask_tensor(_drop_mask((steps, bsz, self.att_odim + self.embedding.embedding_dim), p, dev))
        P = ops.linear(enc_pad, att.mlp_enc.weight, att.mlp_enc.bias)
        Q = ops.linear(enc_pad, att.mlp_o.weight, None)
        w0 = AttLoc.initial_weights(enc_len, frames, dev)
        cell = self.LSTMCell
        logits, ws, pred = ops.decoder_sequence(
            P, Q, self.embedding.weight, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh,
            att.mlp_dec.weight, att.loc_conv.weight, att.mlp_att.weight, att.gvec.weight, att.mlp_o.bias,
            self.output_layer.weight, self.output_layer.bias, w0, opts)
        ws = ws.transpose(0, 1)
        # log_softmax -> gather target (or own prediction) -> label smoothing (model.py:354-366), one kernel each way
        index_lb = tok_out_lb.to(dev) if have_ys else pred                       # [L, B] like the time-major logits
        smooth_on = label_smoothing and self.ls_weight > 0 and self.training
        if smooth_on:
            key = str(dev)
            if key not in self._dist_dev:
                self._dist_dev[key] = self.vlabeldist.to(dev).float().contiguous()
        scale = -1.0 / float(loss_norm * steps) if loss_norm else 1.0
        res = ops.label_logprob(logits, index_lb, self._dist_dev[str(dev)] if smooth_on else None,
                                self.ls_weight if smooth_on else 0.0, with_sum=True, sum_scale=scale, with_argmax=pred is None)
        ys_log_probs, total = res[0], res[1]
        if pred is None:
            pred = res[2]
        prediction = pred.transpose(0, 1)
        ys_log_probs = ys_log_probs.transpose(0, 1)
        # sum of all entries (times the loss's constant), from the same kernel (parallel.local_loss uses it)
        if loss_norm:
            ys_log_probs.fused_loss, ys_log_probs.fused_loss_scale = total, scale
        else:
            ys_log_probs.fused_sum = total
        return logits.transpose(0, 1), ys_log_probs, prediction, ws


class E2E(torch.nn.Module):
    """model.py:408-456."""

    def __init__(self, input_dim, enc_hidden_dim, enc_n_layers, subsample, dropout_rate, dec_hidden_dim, att_dim,
                 conv_channels, conv_kernel_size, att_odim, embedding_dim, output_dim, ls_weight, labeldist,
                 pad=0, bos=1, eos=2):
        super(E2E, self).__init__()
        self.encoder = Encoder(input_dim=input_dim, hidden_dim=enc_hidden_dim, n_layers=enc_n_layers,
                               subsample=subsample, dropout_rate=dropout_rate)
        # one AttLoc shared as self.attention and decoder.attention (duplicate state_dict keys)
        self.attention = AttLoc(encoder_dim=enc_hidden_dim, decoder_dim=dec_hidden_dim, att_dim=att_dim,
                                conv_channels=conv_channels, conv_kernel_size=conv_kernel_size, att_odim=att_odim)
        self.decoder = Decoder(output_dim=output_dim, hidden_dim=dec_hidden_dim, embedding_dim=embedding_dim,
                               attention=self.attention, dropout_rate=dropout_rate, att_odim=att_odim,
                               ls_weight=ls_weight, labeldist=labeldist, bos=bos, eos=eos, pad=pad)

    accepts_loss_norm = True           # (parallel.sup_local_loss: this forward takes the loss's normaliser along)

    def forward(self, data, ilens, ys=None, tf_rate=1.0, max_dec_timesteps=200, sample=False, smooth=False,
                scaling=1.0, label_smoothing=True, total_length=None, olength=None, loss_norm=None):
        if data.is_cuda:
            hb.upload_side_stream_for(data.shape[0] * data.shape[1])       # small uploads leave the compute stream when the GPU is the bottleneck
        enc_h, enc_lens = self.encoder(data, ilens, total_length)
        return self.decoder(enc_h, enc_lens, ys, tf_rate=tf_rate, max_dec_timesteps=max_dec_timesteps, sample=sample,
                            smooth=smooth, scaling=scaling, label_smoothing=label_smoothing, olength=olength,
                            loss_norm=loss_norm)

    def mask_and_cal_loss(self, log_probs, ys, mask=None):
        if mask is None:
            seq_len = [y.size(0) + 1 for y in ys]              # +1 for <EOS>
            mask = _seq_mask(seq_len=seq_len, max_len=log_probs.size(1)).to(log_probs.device)
        else:
            seq_len = [y.size(0) for y in ys]
        return -torch.sum(log_probs * mask) / sum(seq_len)


class LM(torch.nn.Module):
    """The judge: 2-layer LSTM language model (model.py:459-573) on the same fused LSTM kernel."""

    def __init__(self, output_dim, embedding_dim, hidden_dim, dropout_rate, n_layers, bos, eos, pad, ls_weight,
                 labeldist):
        super(LM, self).__init__()
        self.bos, self.eos, self.pad = bos, eos, pad
        self.embedding = torch.nn.Embedding(output_dim, embedding_dim, padding_idx=pad)
        self.LSTM = _LstmWeights(embedding_dim, hidden_dim, num_layers=n_layers, bidirectional=False)
        # re-init as utils.weight_init does for nn.LSTM (utils.py:97-103): orthogonal matrices, normal biases
        for prm in self.LSTM.parameters():
            if prm.dim() >= 2:
                torch.nn.init.orthogonal_(prm.data)
            else:
                torch.nn.init.normal_(prm.data)
        self.output_layer = torch.nn.Linear(hidden_dim, output_dim)
        self.dropout_layer = torch.nn.Dropout(p=dropout_rate)
        self.hidden_dim, self.output_dim = hidden_dim, output_dim
        self.dropout_rate, self.n_layers = dropout_rate, n_layers
        self.ls_weight = ls_weight
        self.labeldist = labeldist
        self._dist_dev = {}
        if labeldist is not None:
            self.vlabeldist = cc(torch.from_numpy(np.array(labeldist, dtype=np.float32)))

    def _run_lstm(self, x_tm, lens_dev):
        """x_tm [T,B,E] -> [T,B,H] through all layers; inter-layer dropout like nn.LSTM(dropout=p)."""
        packs = ops.lstm_pack([self.LSTM.direction_params(l) for l in range(self.n_layers)], 1)    # all layers: one launch
        for l in range(self.n_layers):
            x_tm = ops.lstm_layer(x_tm, lens_dev, None, 1, packed=packs[l])
            if l + 1 < self.n_layers and self.training and self.dropout_rate > 0:
                x_tm = F.dropout(x_tm, self.dropout_rate, True)
        return x_tm

    def forward(self, ys=None, discrete_input=True):
        """-> (ys_log_probs, ys_probs, predictions), each [B,L] (model.py:492-532)."""
        dev = self.embedding.weight.device
        if discrete_input:
            bos = ys[0].new_tensor([self.bos])
            eos = ys[0].new_tensor([self.eos])
            seq_in = [torch.cat([bos, y, eos, eos, eos, eos]) for y in ys]
            seq_out = [torch.cat([y, eos, eos, eos, eos, eos]) for y in ys]
            tok_in = pad_list(seq_in, self.eos).to(dev)
            tok_out = pad_list(seq_out, self.eos).to(dev)
            lens = [int(s.size(0)) for s in seq_in]
        else:
            first = torch.full((ys.size(0), 1), self.bos, dtype=ys.dtype, device=ys.device)
            tok_in = torch.cat([first, ys[:, :-1]], dim=1).to(dev)
            tok_out = ys.to(dev)
            lens = [tok_in.size(1)] * tok_in.size(0)
        eys = self.dropout_layer(self.embedding(tok_in))
        lens_dev = hb_to_device(lens, dev)
        out = self._run_lstm(eys.transpose(0, 1).contiguous(), lens_dev).transpose(0, 1)
        out = self.dropout_layer(out)
        logits = ops.linear(out.contiguous(), self.output_layer.weight, self.output_layer.bias)
        # log_softmax -> gather -> label smoothing (model.py:523-531) on the decoder's kernel (asr_label_logprob_*)
        plain, predictions = ops.label_logprob(logits, tok_out, with_argmax=True)      # log p(target); argmax of the logits
        ys_probs = plain.exp()
        if self.ls_weight > 0 and self.training:
            if str(dev) not in self._dist_dev:
                self._dist_dev[str(dev)] = self.vlabeldist.to(dev).float().contiguous()
            ys_log_probs = ops.label_logprob(logits, tok_out, self._dist_dev[str(dev)], self.ls_weight)
        else:
            ys_log_probs = plain
        return ys_log_probs, ys_probs, predictions

    def zero_state(self, ref, dim=None):
        """model.py:486-490."""
        return ref.new_zeros(self.n_layers, ref.size(0), dim if dim else self.hidden_dim)

    def forward_step(self, emb, dec_z=None, dec_c=None):
        """One step of the stacked LSTM + output layer with carried state (model.py:534-542; decode stage only, no
        autograd): emb [B, 1, E], dec_z / dec_c [n_layers, B, H] or None -> (logit [B, V], dec_z, dec_c).  The products
        run on asr_gemm_f32; inter-layer dropout as nn.LSTM(dropout=p) applies it (training mode only)."""
        with torch.no_grad():
            x = emb.reshape(emb.size(0), -1).contiguous()
            if dec_z is None:
                dec_z, dec_c = self.zero_state(x), self.zero_state(x)
            new_z, new_c = [], []
            for l in range(self.n_layers):
                w_ih, w_hh, b_ih, b_hh = self.LSTM.direction_params(l)
                gates = hb.gemm(x, w_ih, trans_b=True, bias=b_ih)
                hb.gemm(dec_z[l].contiguous(), w_hh, trans_b=True, bias=b_hh, out=gates, accumulate=True)
                gi, gf, gg, go = gates.chunk(4, dim=1)
                cl = torch.sigmoid(gf) * dec_c[l] + torch.sigmoid(gi) * torch.tanh(gg)
                zl = torch.sigmoid(go) * torch.tanh(cl)
                new_z.append(zl)
                new_c.append(cl)
                x = zl
                if l + 1 < self.n_layers and self.training and self.dropout_rate > 0:
                    x = F.dropout(x, self.dropout_rate, True)
            logit = hb.gemm(x.contiguous(), self.output_layer.weight, trans_b=True, bias=self.output_layer.bias)
        return logit, torch.stack(new_z), torch.stack(new_c)

    def decode(self, n_samples=5, sample=False, max_dec_timesteps=500):
        """Free-running generation with carried state (model.py:544-563; lm_validation's samples)."""
        dev = self.embedding.weight.device
        prev = torch.full((n_samples,), self.bos, dtype=torch.long, device=dev)
        dec_z = dec_c = None
        preds = []
        with torch.no_grad():
            for t in range(max_dec_timesteps):
                logit, dec_z, dec_c = self.forward_step(self.embedding(prev).unsqueeze(1), dec_z, dec_c)
                prev = torch.distributions.Categorical(logits=logit).sample() if sample else logit.argmax(-1)
                preds.append(prev)
        return torch.stack(preds, dim=1)

    def mask_and_cal_sum(self, log_probs, ys, mask=None):
        if mask is None:
            seq_len = [y.size(0) + 1 + 4 for y in ys]
            mask = _seq_mask(seq_len=seq_len, max_len=log_probs.size(1)).to(log_probs.device)
        else:
            seq_len = [y.size(0) for y in ys]
        return torch.sum(log_probs * mask) / sum(seq_len)
