"""Pins oracle/asr_oracle.py against fixtures produced by the real reference
(tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

import synth
from oracle import asr_oracle as O

TOL = dict(rtol=2e-5, atol=2e-6)


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name), allow_pickle=False))


def _tiny(golden_dir):
    g = _load(golden_dir, "tiny_e2e.npz")
    cfg = dict(synth.TINY, labeldist=g["labeldist"])
    sd = O.make_leaf_state(synth.e2e_weights(synth.TINY, 11))
    xs, ilens, ys = synth.batch(cfg["input_dim"], cfg["output_dim"], synth.TINY_ILENS, synth.TINY_YLENS, 13)
    return g, cfg, sd, torch.from_numpy(xs), ilens, [torch.from_numpy(y) for y in ys]


def _close(a, b, **kw):
    tol = dict(TOL)
    tol.update(kw)
    np.testing.assert_allclose(a.detach().numpy() if torch.is_tensor(a) else a, b, **tol)


def test_encoder_layers(golden_dir):
    g, cfg, sd, xs, ilens, ys = _tiny(golden_dir)
    taps = {}
    enc_h, lens = O.encoder_forward(sd, xs, ilens, cfg["enc_n_layers"], cfg["subsample"], taps=taps)
    assert lens == g["enc_lens"].tolist()
    for k, v in taps.items():
        _close(v, g["enc_" + k])
    _close(enc_h, g["enc_h"])
    # padded frames equal relu(bias) of the projection
    bias = torch.relu(sd["encoder.enc2.project_layers.1.bias"]).detach().numpy()
    np.testing.assert_allclose(g["enc_h"][2, lens[2]:], np.broadcast_to(bias, g["enc_h"][2, lens[2]:].shape), atol=1e-6)


def test_attention_steps(golden_dir):
    g, cfg, sd, xs, ilens, ys = _tiny(golden_dir)
    enc_h = torch.from_numpy(g["enc_h"])
    lens = g["enc_lens"].tolist()
    st = O.AttState()
    c0, w0 = O.attloc_step(sd, st, enc_h, lens, torch.from_numpy(g["att_z0"]), None)
    c1, w1 = O.attloc_step(sd, st, enc_h, lens, torch.from_numpy(g["att_z1"]), w0)
    _close(c0, g["att_c0"]); _close(w0, g["att_w0"]); _close(c1, g["att_c1"]); _close(w1, g["att_w1"])
    # unmasked softmax: mass on padded frames is non-zero
    assert g["att_w0"][2, lens[2]:].sum() > 1e-3


def test_teacher_forced_loss_and_grads(golden_dir):
    g, cfg, sd, xs, ilens, ys = _tiny(golden_dir)
    np.random.seed(5)
    logits, lp, pred, ws = O.e2e_forward(sd, cfg, xs, ilens, ys, tf_rate=1.0)
    _close(logits, g["tf_logits"]); _close(lp, g["tf_lp"]); _close(ws, g["tf_ws"])
    assert (pred.numpy() == g["tf_pred"]).all()
    loss = -lp.mean()
    _close(loss, g["tf_loss"])
    _close(O.masked_loss(lp, ys), g["tf_masked_loss"])
    names = O.unique_param_names(sd)
    grads = torch.autograd.grad(loss, [sd[n] for n in names])
    for n, gr in zip(names, grads):
        _close(gr, g["grad/" + n], rtol=2e-4, atol=2e-6)


def test_scheduled_sampling_consumes_numpy_rng(golden_dir):
    g, cfg, sd, xs, ilens, ys = _tiny(golden_dir)
    np.random.seed(7)
    logits, lp, pred, _ = O.e2e_forward(sd, cfg, xs, ilens, ys, tf_rate=0.5)
    _close(logits, g["ss_logits"]); _close(lp, g["ss_lp"])
    assert (pred.numpy() == g["ss_pred"]).all()


def test_greedy_smooth_eval(golden_dir):
    g, cfg, sd, xs, ilens, ys = _tiny(golden_dir)
    logits, lp, pred, ws = O.e2e_forward(sd, cfg, xs, ilens, None, max_dec_timesteps=5)
    _close(logits, g["gr_logits"]); _close(lp, g["gr_lp"]); _close(ws, g["gr_ws"])
    assert (pred.numpy() == g["gr_pred"]).all()
    logits, lp, pred, _ = O.e2e_forward(sd, cfg, xs, ilens, None, max_dec_timesteps=5, smooth=True,
                                        scaling=3.0, label_smoothing=False)
    _close(logits, g["sm_logits"]); _close(lp, g["sm_lp"])
    names = O.unique_param_names(sd)
    grads = torch.autograd.grad(-lp.mean(), [sd[n] for n in names])
    for n, gr in zip(names, grads):
        _close(gr, g["smgrad/" + n], rtol=2e-4, atol=2e-6)
    np.random.seed(5)
    _, lp_eval, _, _ = O.e2e_forward(sd, cfg, xs, ilens, ys, training=False)
    _close(lp_eval, g["eval_lp"])


def test_three_optimizer_steps(golden_dir):
    g, cfg, sd, xs, ilens, ys = _tiny(golden_dir)
    names = O.unique_param_names(sd)
    opt = O.AdamAmsgrad(names, lr=5e-4, weight_decay=1e-6)
    for step in range(3):
        np.random.seed(100 + step)
        loss, gnorm, _ = O.sup_train_step(sd, cfg, opt, xs, ilens, ys, max_grad_norm=5.0)
        np.testing.assert_allclose(loss, g["opt_loss%d" % step], rtol=1e-5)
        np.testing.assert_allclose(gnorm, g["opt_gnorm%d" % step], rtol=1e-4)
        if step in (0, 2):
            for n in names:
                _close(sd[n], g["after%d/%s" % (step + 1, n)], rtol=1e-5, atol=1e-6)


def test_clip_branch(golden_dir):
    g, cfg, sd, xs, ilens, ys = _tiny(golden_dir)
    names = O.unique_param_names(sd)
    opt = O.AdamAmsgrad(names, lr=5e-4, weight_decay=1e-6)
    np.random.seed(100)
    O.sup_train_step(sd, cfg, opt, xs, ilens, ys, max_grad_norm=0.05)
    for n in names:
        _close(sd[n], g["clip/" + n], rtol=1e-5, atol=1e-6)


def _lm(golden_dir):
    g = _load(golden_dir, "tiny_lm.npz")
    sd = {k: torch.tensor(v, requires_grad=True) for k, v in synth.lm_weights(synth.TINY_LM, 31).items()}
    ys = [torch.from_numpy(g["ys%d" % i]) for i in range(3)]
    kw = dict(n_layers=2, ls_weight=0.05, labeldist=g["labeldist"])
    return g, sd, ys, kw


def test_lm_forward_modes(golden_dir):
    g, sd, ys, kw = _lm(golden_dir)
    lp, p, pred = O.lm_forward(sd, ys, True, **kw)
    _close(lp, g["d_lp"]); _close(p, g["d_p"])
    valid = O.length_mask([len(y) + 5 for y in ys], lp.shape[1]).numpy().astype(bool)
    assert (pred.numpy()[valid] == g["d_pred"][valid]).all()
    loss = -O.lm_masked_sum(lp, ys)
    _close(loss, g["d_loss"]); _close(O.lm_masked_sum(p, ys), g["d_avg_prob"])
    grads = torch.autograd.grad(loss, list(sd.values()))
    for n, gr in zip(sd, grads):
        _close(gr, g["grad/" + n], rtol=2e-4, atol=2e-6)
    lp_e, _, _ = O.lm_forward(sd, ys, True, training=False, **kw)
    _close(lp_e, g["d_lp_eval"])
    lp, p, pred = O.lm_forward(sd, torch.from_numpy(g["c_ys"]), False, **kw)
    _close(lp, g["c_lp"]); _close(p, g["c_p"])
    assert (pred.numpy() == g["c_pred"]).all()


def test_lm_judge_step(golden_dir):
    g, sd, ys, kw = _lm(golden_dir)
    names = list(sd)
    opt = O.AdamAmsgrad(names, lr=2e-4)
    opt_plain_vmax = True  # amsgrad's running max equals v on the first step
    lp, _, _ = O.lm_forward(sd, ys, True, **kw)
    loss = -O.lm_masked_sum(lp, ys)
    grads = torch.autograd.grad(loss, [sd[n] for n in names])
    clipped, _ = O.clip_global_norm(list(grads), 5.0)
    opt.step(sd, dict(zip(names, clipped)))
    assert opt_plain_vmax
    for n in names:
        _close(sd[n], g["after1/" + n], rtol=1e-5, atol=1e-6)


def test_ssl_losses(golden_dir):
    g = _load(golden_dir, "tiny_ssl.npz")
    t = _load(golden_dir, "tiny_e2e.npz")
    cfg = dict(synth.TINY, labeldist=t["labeldist"])
    sd = O.make_leaf_state(synth.e2e_weights(synth.TINY, 11))
    jsd = {k: torch.tensor(v, requires_grad=True) for k, v in synth.lm_weights(synth.TINY_LM, 31).items()}
    jcfg = dict(n_layers=2, ls_weight=0.05, labeldist=_load(golden_dir, "tiny_lm.npz")["labeldist"])
    xs, ilens, ys = synth.batch(cfg["input_dim"], cfg["output_dim"], synth.TINY_ILENS, synth.TINY_YLENS, 13)
    uxs, uilens, _ = synth.batch(cfg["input_dim"], cfg["output_dim"], [12, 10, 7], [2, 2, 2], 41)
    np.random.seed(9)
    sup, unsup = O.ssl_losses(sd, jsd, cfg, jcfg, torch.from_numpy(xs), ilens, [torch.from_numpy(y) for y in ys],
                              torch.from_numpy(uxs), uilens, float(g["proportion"]))
    _close(sup, g["sup"]); _close(unsup, g["unsup"], rtol=1e-4)
    loss = sup + float(g["unsup_weight"]) * unsup
    names = O.unique_param_names(sd)
    grads = torch.autograd.grad(loss, [sd[n] for n in names])
    for n, gr in zip(names, grads):
        _close(gr, g["grad/" + n], rtol=3e-4, atol=3e-6)


def test_cfg1_shape(golden_dir):
    g = _load(golden_dir, "cfg1.npz")
    cfg = dict(synth.CFG1, labeldist=synth.labeldist(34, 23))
    sd = O.make_leaf_state(synth.e2e_weights(synth.CFG1, 21))
    xs, ilens, ys = synth.batch(80, 34, synth.CFG1_ILENS, synth.CFG1_YLENS, 22)
    np.random.seed(5)
    logits, lp, pred, ws = O.e2e_forward(sd, cfg, torch.from_numpy(xs), ilens, [torch.from_numpy(y) for y in ys])
    _close(logits, g["logits"], rtol=1e-4, atol=1e-5); _close(lp, g["lp"], rtol=1e-4, atol=1e-5)
    _close(ws, g["ws"], rtol=1e-4, atol=1e-6)
    loss = -lp.mean()
    _close(loss, g["loss"], rtol=1e-5)
    names = O.unique_param_names(sd)
    grads = torch.autograd.grad(loss, [sd[n] for n in names])
    for n, gr in zip(names, grads):
        flat = gr.numpy().ravel()
        np.testing.assert_allclose(np.sqrt((flat.astype(np.float64) ** 2).sum()), g["gnorm/" + n], rtol=1e-3)
        np.testing.assert_allclose(flat[:16], g["ghead/" + n], rtol=1e-3, atol=1e-6)


@pytest.mark.parametrize("name,shape", [("cfg2", synth.CFG2_SHAPE), ("cfg5", synth.CFG5_SHAPE)])
def test_big_shapes(golden_dir, name, shape):
    """The oracle at the headline shapes (3x512 model; cfg-2: B=32, T=800; cfg-5: B=8, T=1600) against what the
    reference produced there: forward outputs and every parameter gradient (norm, first / last 16 elements)."""
    g = _load(golden_dir, name + ".npz")
    cfg = dict(synth.CFG2, labeldist=synth.labeldist(34, shape["ldseed"]))
    sd = O.make_leaf_state(synth.e2e_weights(synth.CFG2, shape["wseed"]))
    xs, ilens, ys = synth.ragged_batch(shape["n_utt"], shape["t_max"], 80, 34, shape["bseed"])
    assert ilens == g["ilens"].tolist()
    np.random.seed(5)
    logits, lp, pred, ws = O.e2e_forward(sd, cfg, torch.from_numpy(xs), ilens, [torch.from_numpy(y) for y in ys])
    _close(-lp.mean(), g["loss"], rtol=1e-6)
    _close(lp, g["lp"])
    _close(logits[:, :4], g["logits_head"], atol=2e-5)
    _close(logits[:, -2:], g["logits_tail"], atol=2e-5)
    _close(ws[:, 0], g["ws_first"])
    _close(ws[:, -1], g["ws_last"])
    assert (pred.numpy() == g["pred"]).mean() > 0.999
    names = O.unique_param_names(sd)
    grads = torch.autograd.grad(-lp.mean(), [sd[n] for n in names])
    for n, gr in zip(names, grads):
        flat = gr.numpy().ravel()
        np.testing.assert_allclose(np.sqrt((flat.astype(np.float64) ** 2).sum()), g["gnorm/" + n], rtol=2e-5)
        scale = float(np.abs(flat).max())
        assert np.abs(flat[:16] - g["ghead/" + n]).max() <= 2e-5 * scale + 1e-9, n
        assert np.abs(flat[-16:] - g["gtail/" + n]).max() <= 2e-5 * scale + 1e-9, n
        samp = flat[synth.grad_sample_index(names.index(n), flat.size)]
        assert np.abs(samp - g["gsample/" + n]).max() <= 2e-5 * scale + 1e-9, n


def test_big_ssl_step(golden_dir):
    """The oracle's semi-supervised generator step at the real width (3x512 model + 2x640 judge, 8 + 8 utterances of
    T = 400) against what the reference's solver.py:460-483 arithmetic produced there: the hypothesis of the smooth
    free-running decode, its log-probs, the judge's probabilities, the three losses and every gradient (norm, ends,
    seeded 4 096-element sample)."""
    g = _load(golden_dir, "big_ssl.npz")
    sh = synth.BIG_SSL_SHAPE
    cfg = dict(synth.CFG2, labeldist=synth.labeldist(34, sh["ldseed"]))
    sd = O.make_leaf_state(synth.e2e_weights(synth.CFG2, sh["wseed"]))
    jsd = {k: torch.tensor(v, requires_grad=True) for k, v in synth.lm_weights(synth.CFG_JUDGE, sh["jseed"]).items()}
    jcfg = dict(n_layers=2, ls_weight=0.05, labeldist=synth.labeldist(34, sh["jldseed"]))
    xs, ilens, ys = synth.ragged_batch(sh["n_lab"], sh["t_max"], 80, 34, sh["bseed"])
    uxs, uilens, _ = synth.ragged_batch(sh["n_unlab"], sh["t_max"], 80, 34, sh["ubseed"])
    assert ilens == g["ilens"].tolist() and uilens == g["uilens"].tolist()
    np.random.seed(9)
    sup, unsup = O.ssl_losses(sd, jsd, cfg, jcfg, torch.from_numpy(xs), ilens, [torch.from_numpy(y) for y in ys],
                              torch.from_numpy(uxs), uilens, sh["proportion"], scaling=sh["scaling"])
    _close(sup, g["sup"], rtol=1e-6)
    _close(unsup, g["unsup"], rtol=1e-5)
    loss = sup + sh["unsup_weight"] * unsup
    names = O.unique_param_names(sd)
    grads = torch.autograd.grad(loss, [sd[n] for n in names])
    for i, (n, gr) in enumerate(zip(names, grads)):
        flat = gr.numpy().ravel()
        np.testing.assert_allclose(np.sqrt((flat.astype(np.float64) ** 2).sum()), g["gnorm/" + n], rtol=5e-5)
        scale = float(g["gmax/" + n])
        samp = flat[synth.grad_sample_index(i, flat.size)]
        assert np.abs(samp - g["gsample/" + n]).max() <= 5e-5 * scale + 1e-9, n


def test_text_helpers(golden_dir):
    with open(os.path.join(golden_dir, "text.json")) as f:
        g = json.load(f)
    cut = O.cut_at_eos(g["preds"])
    assert cut == g["cut"]
    hyp = O.ids_to_sentences(cut, g["vocab"], g["non_lang_syms"])
    ref = O.ids_to_sentences(g["refs"], g["vocab"], g["non_lang_syms"])
    assert hyp == g["hyp"] and ref == g["ref"]
    assert abs(O.corpus_cer(hyp, ref) - g["cer"]) < 1e-12
    assert O.length_mask([3, 1, 4], 5).tolist() == g["seq_mask"]
