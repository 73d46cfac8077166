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


# ---- a corpus a model can learn (make_golden.py gen_solver_run; tests regenerate the identical pickles from the seeds)
WSJ_SYMBOLS = (["<PAD>", "<BOS>", "<EOS>"] + [chr(ord("A") + i) for i in range(26)] + ["'", ".", "-", "<space>", "<NOISE>"])
NON_LANG_SYMS = ["<NOISE>", "<PAD>", "<BOS>", "<EOS>"]


def wsj_vocab():
    """The character inventory of the reference's WSJ recipe (preprocess.py:25-36,73 -> V = 34), ids in its order."""
    return {s: i for i, s in enumerate(WSJ_SYMBOLS)}


def learnable_corpus(n, seed, input_dim=80, words=(2, 4), word_len=(1, 4), hold=(6, 10), noise=0.3, proto_seed=4242,
                     p_noise_sym=0.05):
    """{utt: {'feature': f32[T, D], 'token_ids': list}} (dataset.py:46-80's format) in which every token id owns a fixed
    prototype vector (N(0,1), drawn once from `proto_seed`: shared by train / dev / eval) that its frames repeat for
    `hold` frames + N(0, noise) - a monotonic, learnable alignment task.  Transcripts are words of letters joined by
    <space>, now and then a <NOISE> symbol (which CER scoring strips, preprocess.py:81)."""
    vocab = wsj_vocab()
    proto = np.random.RandomState(proto_seed).normal(0, 1, size=(len(vocab), input_dim)).astype(np.float32)
    rs = np.random.RandomState(seed)
    letters = list(range(3, 3 + 26 + 3))
    out = {}
    for i in range(n):
        toks = []
        for w in range(int(rs.randint(words[0], words[1] + 1))):
            if w:
                toks.append(vocab["<space>"])
            if rs.uniform() < p_noise_sym:
                toks += [vocab["<NOISE>"], vocab["<space>"]]
            toks += [int(letters[j]) for j in rs.randint(0, len(letters), size=int(rs.randint(word_len[0], word_len[1] + 1)))]
        frames = []
        for t in toks:
            d = int(rs.randint(hold[0], hold[1] + 1))
            frames.append(proto[t][None, :] + rs.normal(0, noise, size=(d, input_dim)).astype(np.float32))
        out["utt%05d" % i] = dict(feature=np.concatenate(frames, 0).astype(np.float32), token_ids=[int(t) for t in toks])
    return out


# The run both Solvers make over that corpus (make_golden.py gen_solver_run drives the reference's; tests/test_solver_run_gpu.py
# the product's): cfg-1's model (BASELINE configs[0]: 1 x 128 BiLSTM encoder, 320-unit decoder), no dropout (the two sides
# have different dropout generators), no shuffling (the reference never seeds its sampler), a teacher-forcing rate that
# decays, so that the scheduled-sampling draws (model.py:328) and the rule for them (solver.py:414-418) are part of what
# is pinned; then judge pre-training across its learning-rate milestone and semi-supervised iterations with an auxiliary
# weight large enough to move the model.
SOLVER_RUN = dict(
    corpus=dict(train=(400, 101), dev=(224, 102), eval=(16, 103)),           # (utterances, seed)
    numpy_seed=7, model_wseed=311, judge_wseed=312,
    config=dict(labeled_set="train", unlabeled_speech_set="train", unlabeled_text_set="train", dev_set="dev", test_set="eval",
                min_feature_length=4, max_dec_timesteps=30, batch_size=8, shuffle=False, input_dim=80, enc_hidden_dim=128,
                enc_n_layers=1, subsample=[2], dec_hidden_dim=320, att_dim=320, att_odim=320, embedding_dim=128,
                dropout_rate=0.0, dis_dropout_rate=0.0, dis_hidden_dim=128, dis_embedding_dim=64, dis_layers=2,
                epochs=12, learning_rate=0.002, init_tf_rate=1.0, tf_rate_lowerbound=0.8, tf_decay_epochs=8,
                judge_epochs=3, dis_change_learning_rate_epoch=2, d_learning_rate=0.002, lr_gamma=0.2,
                ssl_iterations=40, summary_steps=20, g_learning_rate=0.0005, unsup_weight=0.2, smooth_embedding=True,
                softmax_scaling=3, add_gaussian=False))
# the same supervised run with the reference's default dropout: the two sides draw different masks, so this one is held to a band
SOLVER_RUN_DROPOUT = dict(dropout_rate=0.3, epochs=12)


def write_solver_run_corpus(root, sizes=None):
    """Pickles + vocabulary files of SOLVER_RUN under `root`, in the reference's on-disk format."""
    import os
    import pickle
    for name, (n, seed) in (sizes or SOLVER_RUN["corpus"]).items():
        with open(os.path.join(root, name + ".pkl"), "wb") as f:
            pickle.dump(learnable_corpus(n, seed), f)
    with open(os.path.join(root, "vocab_dict.pkl"), "wb") as f:
        pickle.dump(wsj_vocab(), f)
    with open(os.path.join(root, "non_lang_syms.pkl"), "wb") as f:
        pickle.dump(list(NON_LANG_SYMS), f)


def solver_run_config(base, root, **over):
    """`base` (a config.yaml's 62 keys) with SOLVER_RUN's values and the paths under `root`."""
    import os
    cfg = dict(base)
    cfg.update(logdir=os.path.join(root, "log"), model_dir=root, model_name="m", load_model_path=os.path.join(root, "m"),
               load_judge_path=os.path.join(root, "m"), dataset_root_dir=root, vocab_path=os.path.join(root, "vocab_dict.pkl"),
               non_lang_syms_path=os.path.join(root, "non_lang_syms.pkl"))
    cfg.update(SOLVER_RUN["config"])
    cfg.update(over)
    return cfg


def solver_run_model_cfg(cfg):
    """The constructor arguments e2e_weights / lm_weights need, from a Solver config."""
    V = len(WSJ_SYMBOLS)
    model = {k: cfg[k] for k in ("input_dim", "enc_hidden_dim", "enc_n_layers", "subsample", "dec_hidden_dim", "att_dim",
                                 "conv_channels", "conv_kernel_size", "att_odim", "embedding_dim")}
    model["output_dim"] = V
    judge = dict(output_dim=V, embedding_dim=cfg["dis_embedding_dim"], hidden_dim=cfg["dis_hidden_dim"], n_layers=cfg["dis_layers"])
    return model, judge


# scripted control flow (make_golden.py gen_solver_loops): the CERs / validation losses the stubbed validations return, ties and
# values above the loops' initial bests included
SOLVER_LOOPS = dict(
    corpus=dict(train=(20, 201), dev=(6, 202), eval=(2, 203)),
    config=dict(batch_size=8, epochs=7, init_tf_rate=1.0, tf_rate_lowerbound=0.6, tf_decay_epochs=4, judge_epochs=5,
                dis_change_learning_rate_epoch=3, d_learning_rate=0.002, lr_gamma=0.2, ssl_iterations=7, summary_steps=2,
                g_learning_rate=0.0003, learning_rate=0.002),
    sup_cers=[250.0, 0.9, 0.5, 0.5, 0.7, 0.3, 0.31], judge_val_losses=[150.0, 90.0, 95.0, 80.0, 80.0],
    ssl_cers=[2.5, 1.9, 1.9, 1.2])
