"""Deterministic synthetic weights / batches from numpy's RandomState (MT19937,
stream-stable across numpy versions), so large fixtures need not store weights.

Shapes follow the reference constructors (model.py:59-74,115-123,257-263,409-437,
460-472); values are U(-1/sqrt(fan), 1/sqrt(fan)) like torch's default init but
drawn from numpy so they can be regenerated anywhere.
"""
import numpy as np

TINY = dict(input_dim=8, enc_hidden_dim=16, enc_n_layers=2, subsample=[2, 2], dropout_rate=0.0,
            dec_hidden_dim=16, att_dim=16, conv_channels=2, conv_kernel_size=3, att_odim=16,
            embedding_dim=16, output_dim=9, ls_weight=0.05)
TINY_ILENS = [11, 9, 6]          # odd T exercises the replicate-pad
TINY_YLENS = [4, 3, 2]

CFG1 = dict(input_dim=80, enc_hidden_dim=128, enc_n_layers=1, subsample=[2], dropout_rate=0.0,
            dec_hidden_dim=320, att_dim=320, conv_channels=10, conv_kernel_size=100, att_odim=320,
            embedding_dim=128, output_dim=34, ls_weight=0.05)
CFG1_ILENS = [200, 180, 150, 120]
CFG1_YLENS = [25, 22, 18, 15]

CFG2 = dict(input_dim=80, enc_hidden_dim=512, enc_n_layers=3, subsample=[2, 2, 2], dropout_rate=0.0,
            dec_hidden_dim=512, att_dim=512, conv_channels=10, conv_kernel_size=100, att_odim=512,
            embedding_dim=128, output_dim=34, ls_weight=0.05)
# (batch, frames, weight seed, batch seed, labeldist seed) of the large reference-held fixtures
CFG2_SHAPE = dict(n_utt=32, t_max=800, wseed=99, bseed=1234, ldseed=5)
CFG5_SHAPE = dict(n_utt=8, t_max=1600, wseed=99, bseed=1235, ldseed=5)

# the judge at its config.yaml width (dis_hidden_dim 640, dis_embedding_dim 256, 2 layers) and the semi-supervised step
# fixture at the 3x512 model's own width (make_golden.py gen_big_ssl): 8 labeled + 8 unlabeled utterances of T = 400
CFG_JUDGE = dict(output_dim=34, embedding_dim=256, hidden_dim=640, dropout_rate=0.0, n_layers=2, ls_weight=0.05)
BIG_SSL_SHAPE = dict(n_lab=8, n_unlab=8, t_max=400, wseed=99, jseed=77, bseed=2234, ubseed=2235, ldseed=5, jldseed=6,
                     proportion=0.125, unsup_weight=0.5, scaling=3.0)


def grad_sample_index(i, numel, n=4096):
    """Seeded element sample of parameter number i's gradient (fixtures store the values, tests regenerate the indices)."""
    return np.random.RandomState(7000 + i).randint(0, numel, size=min(n, numel))


TINY_LM = dict(output_dim=9, embedding_dim=16, hidden_dim=16, dropout_rate=0.0, n_layers=2,
               ls_weight=0.05)


def _u(rs, shape, fan):
    k = 1.0 / np.sqrt(fan)
    return rs.uniform(-k, k, size=shape).astype(np.float32)


def e2e_weights(cfg, seed):
    """Reference-keyed state dict (incl. the duplicated attention keys, F9)."""
    rs = np.random.RandomState(seed)
    H, I = cfg["enc_hidden_dim"], cfg["input_dim"]
    sd = {}
    for i in range(cfg["enc_n_layers"]):
        idim = I if i == 0 else H
        p = "encoder.enc2.layers.%d." % i
        for suf in ("", "_reverse"):
            sd[p + "weight_ih_l0" + suf] = _u(rs, (4 * H, idim), H)
            sd[p + "weight_hh_l0" + suf] = _u(rs, (4 * H, H), H)
            sd[p + "bias_ih_l0" + suf] = _u(rs, (4 * H,), H)
            sd[p + "bias_hh_l0" + suf] = _u(rs, (4 * H,), H)
    for i in range(cfg["enc_n_layers"]):
        pd = 4 * H if cfg["subsample"][i] > 1 else 2 * H
        q = "encoder.enc2.project_layers.%d." % i
        sd[q + "weight"] = _u(rs, (H, pd), pd)
        sd[q + "bias"] = _u(rs, (H,), pd)
    A, D, C, K, O = (cfg["att_dim"], cfg["dec_hidden_dim"], cfg["conv_channels"],
                     cfg["conv_kernel_size"], cfg["att_odim"])
    att = {}
    att["mlp_enc.weight"] = _u(rs, (A, H), H)
    att["mlp_enc.bias"] = _u(rs, (A,), H)
    att["mlp_dec.weight"] = _u(rs, (A, D), D)
    att["mlp_att.weight"] = _u(rs, (A, C), C)
    att["loc_conv.weight"] = _u(rs, (C, 1, 1, 2 * K + 1), 2 * K + 1)
    att["gvec.weight"] = _u(rs, (1, A), A)
    att["mlp_o.weight"] = _u(rs, (O, H), H)
    att["mlp_o.bias"] = _u(rs, (O,), H)
    for k, v in att.items():
        sd["attention." + k] = v
    E, V = cfg["embedding_dim"], cfg["output_dim"]
    emb = rs.normal(0, 1, size=(V, E)).astype(np.float32)
    emb[0] = 0.0                                     # padding_idx row
    sd["decoder.embedding.weight"] = emb
    sd["decoder.LSTMCell.weight_ih"] = _u(rs, (4 * D, E + O), D)
    sd["decoder.LSTMCell.weight_hh"] = _u(rs, (4 * D, D), D)
    sd["decoder.LSTMCell.bias_ih"] = _u(rs, (4 * D,), D)
    sd["decoder.LSTMCell.bias_hh"] = _u(rs, (4 * D,), D)
    sd["decoder.output_layer.weight"] = _u(rs, (V, D + O), D + O)
    sd["decoder.output_layer.bias"] = _u(rs, (V,), D + O)
    for k, v in att.items():
        sd["decoder.attention." + k] = v
    return sd


def lm_weights(cfg, seed):
    rs = np.random.RandomState(seed)
    V, E, H = cfg["output_dim"], cfg["embedding_dim"], cfg["hidden_dim"]
    sd = {}
    emb = rs.normal(0, 1, size=(V, E)).astype(np.float32)
    emb[0] = 0.0
    sd["embedding.weight"] = emb
    for l in range(cfg["n_layers"]):
        idim = E if l == 0 else H
        sd["LSTM.weight_ih_l%d" % l] = _u(rs, (4 * H, idim), H)
        sd["LSTM.weight_hh_l%d" % l] = _u(rs, (4 * H, H), H)
        sd["LSTM.bias_ih_l%d" % l] = _u(rs, (4 * H,), H)
        sd["LSTM.bias_hh_l%d" % l] = _u(rs, (4 * H,), H)
    sd["output_layer.weight"] = _u(rs, (V, H), H)
    sd["output_layer.bias"] = _u(rs, (V,), H)
    return sd


def labeldist(V, seed):
    rs = np.random.RandomState(seed)
    d = rs.uniform(0.5, 1.5, size=V)
    d[0] = 0.0
    d[1] = 0.0
    return d / d.sum()


def batch(input_dim, V, ilens, ylens, seed):
    """Collate-shaped batch (dataloader.py:6-12): zero-padded xs, descending ilens."""
    rs = np.random.RandomState(seed)
    xs = np.zeros((len(ilens), max(ilens), input_dim), dtype=np.float32)
    for b, l in enumerate(ilens):
        xs[b, :l] = rs.normal(0, 1, size=(l, input_dim)).astype(np.float32)
    ys = [rs.randint(3, V, size=(n,)).astype(np.int64) for n in ylens]
    return xs, list(ilens), ys



def ragged_batch(n_utt, t_max, input_dim, V, seed):
    """Seeded synthetic batch in collate layout: N(0,1) features, ragged lengths U[0.6T, T] sorted
    descending with the longest pinned to T, label length max(2, 0.125 T_i), ids in [3, V)."""
    rs = np.random.RandomState(seed)
    lens = sorted([int(v) for v in rs.randint(int(0.6 * t_max), t_max + 1, size=n_utt)], reverse=True)
    lens[0] = t_max
    xs = np.zeros((n_utt, t_max, input_dim), dtype=np.float32)
    ys = []
    for b, l in enumerate(lens):
        xs[b, :l] = rs.normal(0, 1, size=(l, input_dim)).astype(np.float32)
        ys.append(rs.randint(3, V, size=(max(2, int(0.125 * l)),)).astype(np.int64))
    ys[0] = rs.randint(3, V, size=(int(0.125 * t_max),)).astype(np.int64)
    return xs, lens, ys
