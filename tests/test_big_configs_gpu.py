"""The headline configurations at their own shapes, against fixtures the REFERENCE produced at those shapes
(tests/golden/make_golden.py gen_big: reference model.py:408-456 + solver.py:375-383 run on CPU in the build
container): cfg-2 = 3x512 encoder / 512 decoder, B=32, T=800 ragged; cfg-5 = same model, B=8, T=1600 (T'=200).
Weights and inputs are regenerated from the synth seeds.  Tolerance: 1e-3 relative fp32 (BASELINE.json north_star).

Both tests also assert, through hip_backend.LAUNCHES, that the persistent XCD-local kernels are what ran: a silent
ASR_E_SHAPE fallback to the per-step kernels fails the test (hb.require_persistent)."""
import os

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
RTOL = 1e-3


def _gpu():
    import __graft_entry__ as entry
    entry.build()
    assert torch.cuda.is_available()
    return torch.device("cuda")


def _rel(got, want):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    want = np.asarray(want)
    return float(np.abs(got - want).max()) / max(1e-30, float(np.abs(want).max()))


def _run_big(golden_dir, name, shape, dec_bwd_persistent=True, RTOL=RTOL):
    dev = _gpu()
    import hip_backend as hb
    hb.persist_clear_abort(dev)
    import model as M
    g = dict(np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False))
    cfg = synth.CFG2
    net = M.E2E(labeldist=synth.labeldist(cfg["output_dim"], shape["ldseed"]), **cfg).to(dev)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.e2e_weights(cfg, shape["wseed"]).items()})
    net.train()
    xs, ilens, ys = synth.ragged_batch(shape["n_utt"], shape["t_max"], cfg["input_dim"], cfg["output_dim"],
                                       shape["bseed"])
    assert ilens == g["ilens"].tolist() and [len(y) for y in ys] == g["ylens"].tolist()
    xs_d, ys_d = torch.from_numpy(xs).to(dev), [torch.from_numpy(y).to(dev) for y in ys]
    hb.LAUNCHES.clear()
    with hb.require_persistent():
        np.random.seed(5)
        enc_h, enc_lens = net.encoder(xs_d, ilens)
        logits, lp, pred, ws = net.decoder(enc_h, enc_lens, ys_d, tf_rate=1.0)
        loss = -lp.mean()
        net.zero_grad()
    if dec_bwd_persistent:
        with hb.require_persistent():
            loss.backward()
    else:
        loss.backward()
    torch.cuda.synchronize()
    assert not hb.persist_aborted(dev), "abort code %d" % hb.persist_abort_code(dev)
    n_layers = cfg["enc_n_layers"]
    want = {"lstm_fwd_persist": n_layers, "lstm_bwd_persist": n_layers, "dec_fwd_persist": 1,
            "dec_bwd_persist" if dec_bwd_persistent else "dec_bwd_step": 1}
    assert dict(hb.LAUNCHES) == want, (dict(hb.LAUNCHES), want)
    # ---- forward
    assert enc_lens == g["enc_lens"].tolist()
    assert _rel(enc_h[0], g["enc_h_b0"]) < RTOL and _rel(enc_h[-1], g["enc_h_blast"]) < RTOL
    assert abs(float(loss.detach()) - float(g["loss"])) <= 1e-4 * abs(float(g["loss"])), (float(loss.detach()), float(g["loss"]))
    assert _rel(lp, g["lp"]) < RTOL
    assert _rel(logits[:, :4], g["logits_head"]) < RTOL and _rel(logits[:, -2:], g["logits_tail"]) < RTOL
    half = ws.size(1) // 2
    for got, key in ((ws[:, 0], "ws_first"), (ws[:, half], "ws_mid"), (ws[:, -1], "ws_last")):
        assert _rel(got, g[key]) < RTOL, key
    agree = float((pred.cpu().numpy() == g["pred"]).mean())
    assert agree > 0.999, "argmax agreement %.5f" % agree          # ties within fp32 noise may flip a handful
    assert abs(float(net.mask_and_cal_loss(lp, ys_d).detach()) - float(g["masked_loss"])) <= 1e-4 * abs(float(g["masked_loss"]))
    # ---- every parameter gradient: norm, and head / tail elements relative to the gradient's own scale
    worst, per_param = 0.0, []
    for pi, (n, p) in enumerate(net.named_parameters()):
        flat = p.grad.detach().cpu().numpy().ravel()
        norm = float(np.sqrt((flat.astype(np.float64) ** 2).sum()))
        assert abs(norm - float(g["gnorm/" + n])) <= RTOL * float(g["gnorm/" + n]), (n, norm, float(g["gnorm/" + n]))
        scale = float(np.abs(flat).max())
        e = max(np.abs(flat[:16] - g["ghead/" + n]).max(), np.abs(flat[-16:] - g["gtail/" + n]).max()) / scale
        # a seeded sample of 4 096 elements all over the tensor: an error confined to the interior of a 2048 x 512 gradient
        # moves neither the norm nor the ends
        samp = flat[synth.grad_sample_index(pi, flat.size)]
        e = max(float(e), float(np.abs(samp - g["gsample/" + n]).max()) / scale)
        worst = max(worst, float(e))
        per_param.append((float(e), n))
        assert e <= RTOL, (n, float(e))
    print("%s [%s]: loss %.6f (ref %.6f), worst gradient element error %.2e (%s), launches %s" % (
        name, hb.arith_name(), float(loss.detach()), float(g["loss"]), worst,
        ", ".join("%s %.1e" % (n.replace("encoder.enc2.", ""), e) for e, n in sorted(per_param, reverse=True)[:4]), dict(hb.LAUNCHES)))


# The default arithmetic (bf16x6: fp32-equivalent products on the bf16 MFMA) AND the exact fp32-input MFMA kernels are
# both held to the reference at the headline shapes at the north-star gate, 1e-3 relative.  bf16x3 (round 2's default: 16
# significand bits per operand) does NOT meet that gate once the gradients are sampled all over each tensor: at cfg-5
# (1 600-step recurrences) the layer-0 dW_ih sample is 2.2e-3 off - round 2's fixtures (norm + first / last 16 elements)
# could not see it.  It stays selectable and is held to 5e-3 here, which is why it is not the default.
ARITHS = [("bf16x6", RTOL), ("f32", RTOL), ("bf16x3", 5e-3)]


@pytest.mark.parametrize("arith,gate", ARITHS)
def test_cfg2_against_golden(golden_dir, arith, gate):
    """cfg-2 (BASELINE.json configs[1], the bench workload) end to end on the persistent kernels vs the reference."""
    import hip_backend as hb
    _gpu()
    with hb.arith(arith):
        _run_big(golden_dir, "cfg2", synth.CFG2_SHAPE, RTOL=gate)


@pytest.mark.parametrize("arith,gate", ARITHS)
def test_cfg5_against_golden(golden_dir, arith, gate):
    """cfg-5 (configs[4]: 80x1600 frames, batch 8, T'=200, L+1=201) vs the reference: 4-row LSTM groups, both decoder
    kernels in the T' <= 256 geometry (2 utterances per XCD group)."""
    import hip_backend as hb
    _gpu()
    with hb.arith(arith):
        _run_big(golden_dir, "cfg5", synth.CFG5_SHAPE, RTOL=gate)
