"""Generate golden fixtures by running the REAL reference (read-only at
/root/reference) in the build container.  Only data (inputs + expected outputs)
is written; no reference source travels.  Run from anywhere:

    python tests/golden/make_golden.py

Absent third-party modules (tensorboardX, editdistance — plain
ModuleNotFoundError here, no permission issue) are stubbed before import, as
SURVEY 8c describes.  Weights/inputs come from tests/golden/synth.py.
"""
import copy
import json
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np
import torch

import synth


def _stub_modules():
    tb = types.ModuleType("tensorboardX")

    class SummaryWriter:
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

        def add_text(self, *a, **k):
            pass

    tb.SummaryWriter = SummaryWriter
    sys.modules["tensorboardX"] = tb
    ed = types.ModuleType("editdistance")

    def _eval(a, b):
        prev = list(range(len(b) + 1))
        for i, x in enumerate(a, 1):
            cur = [i]
            for j, y in enumerate(b, 1):
                cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
            prev = cur
        return prev[-1]

    ed.eval = _eval
    sys.modules["editdistance"] = ed


_stub_modules()
sys.path.insert(0, "/root/reference")
import model as ref_model   # noqa: E402
import utils as ref_utils   # noqa: E402


def load_sd(module, arrays):
    module.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in arrays.items()})


def npy(t):
    return t.detach().cpu().numpy().copy()


def build_e2e(cfg, wseed, ldseed):
    ld = synth.labeldist(cfg["output_dim"], ldseed)
    m = ref_model.E2E(labeldist=ld, **cfg)
    load_sd(m, synth.e2e_weights(cfg, wseed))
    m.train()
    return m, ld


def to_t(xs, ys):
    return torch.from_numpy(xs), [torch.from_numpy(y) for y in ys]


def named_grads(m):
    return {n: npy(p.grad) for n, p in m.named_parameters()}


def gen_tiny_e2e():
    cfg = synth.TINY
    m, ld = build_e2e(cfg, 11, 12)
    xs_np, ilens, ys_np = synth.batch(cfg["input_dim"], cfg["output_dim"], synth.TINY_ILENS,
                                      synth.TINY_YLENS, 13)
    xs, ys = to_t(xs_np, ys_np)
    out = {"labeldist": ld}
    # ---- encoder intermediates via hooks (model.py:80-81,93)
    taps = {}
    hooks = []
    for i, layer in enumerate(m.encoder.enc2.layers):
        def _h(mod, inp, res, i=i):
            padded, _ = torch.nn.utils.rnn.pad_packed_sequence(res[0], batch_first=True)
            taps["lstm%d" % i] = npy(padded)
        hooks.append(layer.register_forward_hook(_h))
    for i, proj in enumerate(m.encoder.enc2.project_layers):
        def _p(mod, inp, res, i=i):
            taps["cat%d" % i] = npy(inp[0])
            taps["proj%d" % i] = npy(torch.relu(res))
        hooks.append(proj.register_forward_hook(_p))
    enc_h, enc_lens = m.encoder(xs, ilens)
    for h in hooks:
        h.remove()
    for k, v in taps.items():
        out["enc_" + k] = v
    out["enc_h"] = npy(enc_h)
    out["enc_lens"] = np.asarray(enc_lens)
    # ---- two attention steps (model.py:139-173)
    rs = np.random.RandomState(14)
    z0 = torch.from_numpy(rs.normal(0, 1, (len(ilens), cfg["dec_hidden_dim"])).astype(np.float32))
    z1 = torch.from_numpy(rs.normal(0, 1, (len(ilens), cfg["dec_hidden_dim"])).astype(np.float32))
    m.attention.reset()
    c0, w0 = m.attention(enc_h, enc_lens, z0, None)
    c1, w1 = m.attention(enc_h, enc_lens, z1, w0)
    m.attention.reset()
    out.update(att_z0=npy(z0), att_z1=npy(z1), att_c0=npy(c0), att_w0=npy(w0), att_c1=npy(c1),
               att_w1=npy(w1))
    # ---- teacher-forced forward, loss, grads (solver.py:375-383)
    np.random.seed(5)
    logits, lp, pred, ws = m(xs, ilens, ys, tf_rate=1.0)
    loss = -torch.mean(lp)
    m.zero_grad()
    loss.backward()
    out.update(tf_logits=npy(logits), tf_lp=npy(lp), tf_pred=npy(pred), tf_ws=npy(ws),
               tf_loss=npy(loss), tf_masked_loss=npy(m.mask_and_cal_loss(lp, ys)))
    for n, g in named_grads(m).items():
        out["grad/" + n] = g
    # ---- scheduled sampling tf_rate=0.5 (numpy RNG consumed per step, F7)
    np.random.seed(7)
    logits, lp, pred, _ = m(xs, ilens, ys, tf_rate=0.5)
    out.update(ss_logits=npy(logits), ss_lp=npy(lp), ss_pred=npy(pred))
    # ---- greedy and smooth free-running decodes (model.py:330-341)
    logits, lp, pred, ws = m(xs, ilens, ys=None, max_dec_timesteps=5)
    out.update(gr_logits=npy(logits), gr_lp=npy(lp), gr_pred=npy(pred), gr_ws=npy(ws))
    logits, lp, pred, _ = m(xs, ilens, ys=None, max_dec_timesteps=5, smooth=True, scaling=3.0,
                            label_smoothing=False)
    m.zero_grad()
    (-lp.mean()).backward()
    out.update(sm_logits=npy(logits), sm_lp=npy(lp), sm_pred=npy(pred))
    for n, g in named_grads(m).items():
        out["smgrad/" + n] = g
    # ---- eval mode (no label smoothing, F8)
    m.eval()
    np.random.seed(5)
    _, lp_eval, _, _ = m(xs, ilens, ys)
    out["eval_lp"] = npy(lp_eval)
    m.train()
    # ---- 3 optimiser steps (solver.py:152-153,382-385)
    m2, _ = build_e2e(cfg, 11, 12)
    opt = torch.optim.Adam(m2.parameters(), lr=5e-4, weight_decay=1e-6, amsgrad=True)
    for step in range(3):
        np.random.seed(100 + step)
        _, lp, _, _ = m2(xs, ilens, ys, tf_rate=1.0)
        loss = -torch.mean(lp)
        opt.zero_grad()
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(m2.parameters(), max_norm=5)
        opt.step()
        out["opt_loss%d" % step] = npy(loss)
        out["opt_gnorm%d" % step] = npy(gn)
        if step in (0, 2):
            for n, p in m2.named_parameters():
                out["after%d/%s" % (step + 1, n)] = npy(p)
    # small-clip variant so the clip branch is actually taken
    m3, _ = build_e2e(cfg, 11, 12)
    opt = torch.optim.Adam(m3.parameters(), lr=5e-4, weight_decay=1e-6, amsgrad=True)
    np.random.seed(100)
    _, lp, _, _ = m3(xs, ilens, ys, tf_rate=1.0)
    opt.zero_grad()
    (-torch.mean(lp)).backward()
    torch.nn.utils.clip_grad_norm_(m3.parameters(), max_norm=0.05)
    opt.step()
    for n, p in m3.named_parameters():
        out["clip/%s" % n] = npy(p)
    np.savez_compressed(os.path.join(HERE, "tiny_e2e.npz"), **out)
    print("tiny_e2e: %d arrays, loss %.6f" % (len(out), float(out["tf_loss"])))


def build_lm(cfg, wseed, ldseed):
    ld = synth.labeldist(cfg["output_dim"], ldseed)
    lm = ref_model.LM(bos=1, eos=2, pad=0, labeldist=ld, **cfg)
    load_sd(lm, synth.lm_weights(cfg, wseed))
    lm.train()
    return lm, ld


def gen_tiny_lm():
    cfg = synth.TINY_LM
    lm, ld = build_lm(cfg, 31, 32)
    rs = np.random.RandomState(33)
    ys_np = [rs.randint(3, cfg["output_dim"], size=(n,)).astype(np.int64) for n in (6, 4, 3)]
    ys = [torch.from_numpy(y) for y in ys_np]
    out = {"labeldist": ld}
    for i, y in enumerate(ys_np):
        out["ys%d" % i] = y
    lp, p, pred = lm(ys, discrete_input=True)
    loss = -lm.mask_and_cal_sum(lp, ys)
    lm.zero_grad()
    loss.backward()
    out.update(d_lp=npy(lp), d_p=npy(p), d_pred=npy(pred), d_loss=npy(loss),
               d_avg_prob=npy(lm.mask_and_cal_sum(p, ys)))
    for n, q in lm.named_parameters():
        out["grad/" + n] = npy(q.grad)
    lm.eval()
    lp_e, _, _ = lm(ys, discrete_input=True)
    out["d_lp_eval"] = npy(lp_e)
    lm.train()
    dense = torch.from_numpy(rs.randint(2, cfg["output_dim"], size=(3, 7)).astype(np.int64))
    lp, p, pred = lm(dense, discrete_input=False)
    out.update(c_ys=npy(dense), c_lp=npy(lp), c_p=npy(p), c_pred=npy(pred))
    # one judge optimiser step (solver.py:171-173,288-297)
    lm2, _ = build_lm(cfg, 31, 32)
    opt = torch.optim.Adam(lm2.parameters(), lr=2e-4)
    lp, _, _ = lm2(ys, discrete_input=True)
    loss = -lm2.mask_and_cal_sum(lp, ys)
    opt.zero_grad()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(lm2.parameters(), max_norm=5)
    opt.step()
    for n, q in lm2.named_parameters():
        out["after1/" + n] = npy(q)
    np.savez_compressed(os.path.join(HERE, "tiny_lm.npz"), **out)
    print("tiny_lm: %d arrays, loss %.6f" % (len(out), float(out["d_loss"])))


def gen_tiny_ssl():
    """Loss assembly of solver.py:465-483 with the tiny model + tiny judge."""
    cfg = synth.TINY
    m, _ = build_e2e(cfg, 11, 12)
    lm, _ = build_lm(synth.TINY_LM, 31, 32)
    xs_np, ilens, ys_np = synth.batch(cfg["input_dim"], cfg["output_dim"], synth.TINY_ILENS,
                                      synth.TINY_YLENS, 13)
    uxs_np, uilens, _ = synth.batch(cfg["input_dim"], cfg["output_dim"], [12, 10, 7], [2, 2, 2], 41)
    xs, ys = to_t(xs_np, ys_np)
    uxs = torch.from_numpy(uxs_np)
    proportion = 0.5
    _, u_lp, u_pred, _ = m(uxs, uilens, ys=None, sample=False, label_smoothing=False,
                           max_dec_timesteps=int(uxs.size(1) * proportion), smooth=True, scaling=3)
    _, lm_p, _ = lm(ys=u_pred, discrete_input=False)
    mask = (u_pred != 2).float()
    unsup = -torch.sum(lm_p * u_lp * mask) / torch.sum(mask)
    np.random.seed(9)
    _, l_lp, _, _ = m(xs, ilens, ys=ys, tf_rate=1.0, sample=False)
    sup = -torch.mean(l_lp)
    loss = sup + 0.5 * unsup
    m.zero_grad()
    lm.zero_grad()
    loss.backward()
    out = dict(unsup=npy(unsup), sup=npy(sup), loss=npy(loss), u_pred=npy(u_pred), u_lp=npy(u_lp),
               lm_p=npy(lm_p), unsup_weight=np.float32(0.5), proportion=np.float32(proportion))
    for n, g in named_grads(m).items():
        out["grad/" + n] = g
    np.savez_compressed(os.path.join(HERE, "tiny_ssl.npz"), **out)
    print("tiny_ssl: sup %.6f unsup %.6f" % (float(sup), float(unsup)))


def gen_cfg1():
    cfg = synth.CFG1
    m, ld = build_e2e(cfg, 21, 23)
    xs_np, ilens, ys_np = synth.batch(cfg["input_dim"], cfg["output_dim"], synth.CFG1_ILENS,
                                      synth.CFG1_YLENS, 22)
    xs, ys = to_t(xs_np, ys_np)
    np.random.seed(5)
    logits, lp, pred, ws = m(xs, ilens, ys, tf_rate=1.0)
    loss = -torch.mean(lp)
    m.zero_grad()
    loss.backward()
    out = dict(logits=npy(logits), lp=npy(lp), pred=npy(pred), ws=npy(ws), loss=npy(loss))
    for n, p in m.named_parameters():
        g = npy(p.grad).ravel()
        out["gnorm/" + n] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
        out["ghead/" + n] = g[:16].copy()
    np.savez_compressed(os.path.join(HERE, "cfg1.npz"), **out)
    print("cfg1: loss %.6f" % float(loss))


def gen_big(name, shape):
    """cfg-2 / cfg-5 shapes held by the reference itself (model.py:408-456, solver.py:375-383): the 3x512 model,
    ragged batch, dropout 0, tf_rate 1.  Weights and inputs are regenerated from synth seeds by the tests; stored are
    the loss, the log-probs, slices of logits / attention weights / encoder output and per-parameter gradient
    norms + 16-element heads (as gen_cfg1 does)."""
    import time
    cfg = synth.CFG2
    m, ld = build_e2e(cfg, shape["wseed"], shape["ldseed"])
    xs_np, ilens, ys_np = synth.ragged_batch(shape["n_utt"], shape["t_max"], cfg["input_dim"], cfg["output_dim"],
                                             shape["bseed"])
    xs, ys = to_t(xs_np, ys_np)
    np.random.seed(5)
    t0 = time.time()
    enc_h, enc_lens = m.encoder(xs, ilens)
    logits, lp, pred, ws = m.decoder(enc_h, enc_lens, ys, tf_rate=1.0)
    loss = -torch.mean(lp)
    m.zero_grad()
    loss.backward()
    print("%s: reference forward+backward %.1f s" % (name, time.time() - t0))
    out = dict(ilens=np.asarray(ilens), ylens=np.asarray([len(y) for y in ys_np]), loss=npy(loss), lp=npy(lp),
               pred=npy(pred), logits_head=npy(logits[:, :4]), logits_tail=npy(logits[:, -2:]),
               ws_first=npy(ws[:, 0]), ws_mid=npy(ws[:, ws.size(1) // 2]), ws_last=npy(ws[:, -1]),
               enc_h_b0=npy(enc_h[0]), enc_h_blast=npy(enc_h[-1]), enc_lens=np.asarray(enc_lens),
               masked_loss=npy(m.mask_and_cal_loss(lp, ys)))
    _store_grads(out, m)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("%s: loss %.6f" % (name, float(loss)))


def _store_grads(out, m):
    """Per parameter: gradient norm, first / last 16 elements and a seeded sample of 4 096 elements (synth.grad_sample_index:
    an error confined to the interior of a large gradient moves neither the norm nor the ends)."""
    for i, (n, p) in enumerate(m.named_parameters()):
        g = npy(p.grad).ravel()
        out["gnorm/" + n] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
        out["ghead/" + n] = g[:16].copy()
        out["gtail/" + n] = g[-16:].copy()
        out["gsample/" + n] = g[synth.grad_sample_index(i, g.size)].copy()
        out["gmax/" + n] = np.float32(np.abs(g).max())


def gen_big_ssl():
    """The semi-supervised generator step at the real width, held by the reference: solver.py:460-483's arithmetic with
    the 3x512 model and the 2x640 judge (dis_dropout_rate 0), 8 unlabeled + 8 labeled utterances of T = 400:
    unlabeled greedy smooth-embedding decode with gradients (int(T * proportion) = 50 steps), judge probabilities of the
    hypothesis, unsup = -sum(p_LM * log p * [pred != EOS]) / sum([pred != EOS]), labeled teacher-forced pass,
    loss = sup + unsup_weight * unsup, backward.  Stored: the three losses, the hypothesis, its log-probs, the judge's
    probabilities, and every generator gradient as norm + ends + seeded sample."""
    import time
    sh = synth.BIG_SSL_SHAPE
    cfg = synth.CFG2
    m, _ = build_e2e(cfg, sh["wseed"], sh["ldseed"])
    lm, _ = build_lm(synth.CFG_JUDGE, sh["jseed"], sh["jldseed"])
    xs_np, ilens, ys_np = synth.ragged_batch(sh["n_lab"], sh["t_max"], cfg["input_dim"], cfg["output_dim"], sh["bseed"])
    uxs_np, uilens, _ = synth.ragged_batch(sh["n_unlab"], sh["t_max"], cfg["input_dim"], cfg["output_dim"], sh["ubseed"])
    xs, ys = to_t(xs_np, ys_np)
    uxs = torch.from_numpy(uxs_np)
    t0 = time.time()
    u_logits, u_lp, u_pred, _ = m(uxs, uilens, ys=None, sample=False, label_smoothing=False,
                                  max_dec_timesteps=int(uxs.size(1) * sh["proportion"]), smooth=True, scaling=sh["scaling"])
    _, lm_p, _ = lm(ys=u_pred, discrete_input=False)
    mask = (u_pred != 2).float()
    unsup = -torch.sum(lm_p * u_lp * mask) / torch.sum(mask)
    np.random.seed(9)
    _, l_lp, _, _ = m(xs, ilens, ys=ys, tf_rate=1.0, sample=False)
    sup = -torch.mean(l_lp)
    loss = sup + sh["unsup_weight"] * unsup
    m.zero_grad()
    lm.zero_grad()
    loss.backward()
    print("big_ssl: reference step %.1f s" % (time.time() - t0))
    top2 = torch.topk(u_logits, 2, dim=2).values
    margin = float((top2[..., 0] - top2[..., 1]).min())
    out = dict(unsup=npy(unsup), sup=npy(sup), loss=npy(loss), u_pred=npy(u_pred), u_lp=npy(u_lp), lm_p=npy(lm_p),
               n_hyp_tokens=np.float32(mask.sum()), min_top2_margin=np.float32(margin),
               ilens=np.asarray(ilens), uilens=np.asarray(uilens))
    _store_grads(out, m)
    np.savez_compressed(os.path.join(HERE, "big_ssl.npz"), **out)
    print("big_ssl: sup %.6f unsup %.6f, %d hypothesis tokens, smallest top-2 logit margin %.3e"
          % (float(sup), float(unsup), int(mask.sum()), margin))


def gen_tiny_opt():
    """torch.optim.Adam's own state_dict() after step 1 of the tiny model (solver.py:38-46 writes exactly this to
    `.opt`), so that loading a reference optimiser checkpoint into the flat optimiser can be tested: resuming from it
    and running steps 2-3 must land on tiny_e2e.npz's after3/* weights."""
    cfg = synth.TINY
    m, _ = build_e2e(cfg, 11, 12)
    xs_np, ilens, ys_np = synth.batch(cfg["input_dim"], cfg["output_dim"], synth.TINY_ILENS, synth.TINY_YLENS, 13)
    xs, ys = to_t(xs_np, ys_np)
    opt = torch.optim.Adam(m.parameters(), lr=5e-4, weight_decay=1e-6, amsgrad=True)
    np.random.seed(100)
    _, lp, _, _ = m(xs, ilens, ys, tf_rate=1.0)
    opt.zero_grad()
    (-torch.mean(lp)).backward()
    torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=5)
    opt.step()
    sd = opt.state_dict()
    out = {}
    names = [n for n, _ in m.named_parameters()]
    out["param_order"] = np.asarray(names)
    grp = sd["param_groups"][0]
    out["group/params"] = np.asarray(grp["params"])
    for k in ("lr", "eps", "weight_decay"):
        out["group/" + k] = np.float64(grp[k])
    out["group/betas"] = np.asarray(grp["betas"], dtype=np.float64)
    out["group/amsgrad"] = np.asarray(bool(grp["amsgrad"]))
    for i, ent in sd["state"].items():
        for k, v in ent.items():
            out["state/%d/%s" % (i, k)] = npy(v) if torch.is_tensor(v) else np.asarray(v)
    for n, p in m.named_parameters():
        out["after1/" + n] = npy(p)
    np.savez_compressed(os.path.join(HERE, "tiny_opt.npz"), **out)
    print("tiny_opt: %d arrays, %d state entries" % (len(out), len(sd["state"])))


def _tensor_norms(module):
    return {n: float(np.sqrt((npy(v).astype(np.float64) ** 2).sum())) for n, v in module.state_dict().items()}


def _ulp_perturbed(arrays, seed):
    """Every weight moved by ONE unit in its last place, up or down (seeded): the smallest perturbation fp32 weights can have."""
    rs = np.random.RandomState(seed)
    out = {}
    for k in sorted(arrays):
        v = np.asarray(arrays[k], dtype=np.float32)
        out[k] = np.where(rs.randint(0, 2, size=v.shape) == 1, np.nextafter(v, np.float32(np.inf)), np.nextafter(v, np.float32(-np.inf))).astype(np.float32)
    for k in list(out):                                  # (the attention module's weights are stored twice, F9: one object)
        if k.startswith("decoder.attention."):
            out[k] = out[k[len("decoder."):]]
    return out


def _rel_perturbed(arrays, rel, seed):
    """Every weight times (1 + rel * N(0, 1)) (seeded): what an implementation whose every product rounds differently - another
    summation order, another split of the operands - amounts to after its first step."""
    rs = np.random.RandomState(seed)
    out = {k: (np.asarray(arrays[k], dtype=np.float32) * (1.0 + rel * rs.normal(0, 1, size=np.shape(arrays[k])))).astype(np.float32)
           for k in sorted(arrays)}
    for k in list(out):
        if k.startswith("decoder.attention."):
            out[k] = out[k[len("decoder."):]]
    return out


def _solver_run(over, stages, ulp_seed=None, rel=None, grad_noise=None):
    """Drive the reference's own Solver (solver.py:13-565; main.py cannot be used, F10) over synth.SOLVER_RUN and record what
    its loops produce.  The methods are wrapped from outside to note their return values; nothing of them is restated."""
    import contextlib
    import io
    import tempfile
    import time
    import yaml
    import solver as ref_solver
    run = synth.SOLVER_RUN
    with open("/root/reference/config.yaml") as f:
        base = yaml.safe_load(f)
    rec = dict(sup=[], judge=[], ssl_steps=[], ssl_summaries=[])
    with tempfile.TemporaryDirectory() as root:
        synth.write_solver_run_corpus(root)
        cfg = synth.solver_run_config(base, root, **over)
        cwd = os.getcwd()
        os.chdir(root)                                   # Solver.test writes {test_set}.txt into the working directory
        try:
            torch.manual_seed(0)
            np.random.seed(run["numpy_seed"])            # the reference seeds nothing (F7): the caller does
            with contextlib.redirect_stdout(io.StringIO()):
                s = ref_solver.Solver(cfg)
            mcfg, jcfg = synth.solver_run_model_cfg(cfg)
            mw, jw = synth.e2e_weights(mcfg, run["model_wseed"]), synth.lm_weights(jcfg, run["judge_wseed"])
            if ulp_seed is not None:
                mw, jw = _ulp_perturbed(mw, ulp_seed), _ulp_perturbed(jw, ulp_seed + 1)
            if rel is not None:
                mw, jw = _rel_perturbed(mw, rel[0], rel[1]), _rel_perturbed(jw, rel[0], rel[1] + 1)
            load_sd(s.model, mw)
            load_sd(s.judge, jw)
            rec.update(proportion=float(s.proportion), labeldist=[float(v) for v in s.labeldist],
                       unlab_labeldist=[float(v) for v in s.unlab_labeldist], steps_per_epoch=len(s.train_lab_loader),
                       judge_steps_per_epoch=len(s.train_unlab_y_loader), dev_batches=len(s.dev_loader))
            t0 = time.time()
            cur = {}
            real_clip = torch.nn.utils.clip_grad_norm_
            if grad_noise is not None:
                # every gradient element of every step off by grad_noise[0] x (the largest element of its tensor) x N(0, 1):
                # what an implementation with its own rounding in every operation looks like to the optimiser (the kernels
                # of the hot path are held to ~1e-6 of a tensor's scale; the parity gate is 1e-3)
                gen = torch.Generator().manual_seed(grad_noise[1])

                def noisy_clip(params, max_norm, *a, **k):
                    params = list(params)
                    for q in params:
                        if q.grad is not None:
                            q.grad.add_(torch.randn(q.grad.shape, generator=gen) * (grad_noise[0] * float(q.grad.abs().max())))
                    return real_clip(params, max_norm, *a, **k)
                torch.nn.utils.clip_grad_norm_ = noisy_clip
            real = dict(epoch=s.sup_train_one_epoch, val=s.validation, lmval=s.lm_validation,
                        jit=s.judge_train_one_iteration, git=s.gen_train_one_iteration)

            def epoch(e, tf_rate):
                cur.update(epoch=int(e), tf_rate=float(tf_rate))
                cur["train_loss"] = float(real["epoch"](e, tf_rate))
                return cur["train_loss"]

            def val():
                out = real["val"]()
                item = dict(cur, val_loss=float(out[0]), cer=float(out[1]), hyps=list(out[2]), refs=list(out[3]))
                (rec["ssl_summaries"] if cur.get("stage") == "ssl" else rec["sup"]).append(item)
                sys.stderr.write("  %s val_loss %.4f CER %.4f (%.0f s)\n" % (cur.get("stage", "sup"), out[0], out[1], time.time() - t0))
                return out

            def lmval():
                out = real["lmval"]()
                rec["judge"].append(dict(val_loss=float(out[0]), losses=cur.pop("jlosses", []), probs=cur.pop("jprobs", [])))
                return out

            def jit(ys):
                meta = real["jit"](ys)
                cur.setdefault("jlosses", []).append(float(meta["loss"]))
                cur.setdefault("jprobs", []).append(float(meta["avg_prob"]))
                return meta

            def git(*a):
                meta = real["git"](*a)
                rec["ssl_steps"].append({k: float(v) for k, v in meta.items()})
                return meta

            s.sup_train_one_epoch, s.validation, s.lm_validation = epoch, val, lmval
            s.judge_train_one_iteration, s.gen_train_one_iteration = jit, git
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(sys.stderr):
                import warnings
                warnings.simplefilter("ignore")
                _, best_cer = s.sup_pretrain()
                rec["sup_best_cer"] = float(best_cer)
                rec["sup_final_norms"] = _tensor_norms(s.model)
                if "judge" in stages:
                    s.judge_pretrain()
                    rec["judge_final_norms"] = _tensor_norms(s.judge)
                if "ssl" in stages:
                    cur.clear()
                    cur["stage"] = "ssl"
                    s.ssl_train()
                    rec["ssl_final_norms"] = _tensor_norms(s.model)
                if "test" in stages:
                    rec["test_cer"] = float(s.test(state_dict=s.model.state_dict()))
                    with open(cfg["test_set"] + ".txt") as f:
                        rec["test_hyps"] = f.read().splitlines()
        finally:
            os.chdir(cwd)
            if grad_noise is not None:
                torch.nn.utils.clip_grad_norm_ = real_clip
    for item in rec["sup"] + rec["ssl_summaries"]:
        item.pop("stage", None)
    rec["seconds"] = round(time.time() - t0, 1)
    return rec


def _spread(primary, others, key):
    """Per entry of `key` (the epochs / summaries): what the reference's OWN runs at other thread counts made of it."""
    out = []
    for i, item in enumerate(primary[key]):
        alt = [o[key][i] for o in others]
        out.append(dict(cer=[a["cer"] for a in alt], val_loss=[a["val_loss"] for a in alt],
                        train_loss=[a.get("train_loss") for a in alt],
                        same_hyps=[sum(h1 == h2 for h1, h2 in zip(item["hyps"], a["hyps"])) for a in alt]))
    return out


def gen_solver_run():
    """The reference's training LOOPS, held by the reference: Solver.sup_pretrain (tf-rate schedule, per-epoch validation ->
    CER, best-CER checkpointing: solver.py:395-458, 360-393, 212-242), judge_pretrain across its learning-rate milestone
    (303-358), ssl_train with two summary periods (497-565, 460-495) and test (244-286), on the learnable corpus of
    synth.SOLVER_RUN.  Stored per epoch / summary: train loss, dev loss, CER, every hypothesis and reference sentence; per
    judge epoch the step losses and the validation loss; per semi-supervised iteration the three losses; tensor norms of
    the weights each stage ends with.

    A training run is a chaotic system and the reference is its own witness: the SAME run at 4, 2 and 1 host threads (other
    summation orders inside some of torch's CPU kernels, nothing else) leaves the primary's trajectory after about 100 steps;
    and the same run with every initial weight moved by ONE unit in its last place - what any implementation with another
    rounding in every product amounts to from the first step on - leaves it sooner.  The fixture therefore carries, per
    epoch, what those runs produced (`spread`, in the order of threads.spread_order): where they agree a port must agree
    too, where they differ the reference's own spread is the resolution at which any implementation can be compared.

    `ssl_early`: ssl_train behind ONE supervised epoch (the runs still agree there), 30 iterations with summaries every 10 - the
    semi-supervised loop inside the window in which the reference agrees with itself.  (Straight from the synthetic weights
    the reference's own loop divides by zero: every hypothesis token is <EOS>, sum(mask) = 0, solver.py:477-478.)
    The last run repeats the supervised stage with dropout 0.3."""
    EARLY = dict(epochs=1, ssl_iterations=30, summary_steps=10)
    threads = [int(t) for t in os.environ.get("GOLDEN_SPREAD_THREADS", "4,2,1").split(",") if t]
    primary_threads = torch.get_num_threads()
    rec = _solver_run({}, ("judge", "ssl", "test"))
    cold = _solver_run(EARLY, ("ssl",))
    others, cold_others = [], []
    for t in threads:
        torch.set_num_threads(t)
        others.append(_solver_run({}, ("judge", "ssl")))
        cold_others.append(_solver_run(EARLY, ("ssl",)))
    torch.set_num_threads(primary_threads)
    ulps = [int(u) for u in os.environ.get("GOLDEN_SPREAD_ULP_SEEDS", "901,902,903").split(",") if u]
    for u in ulps:                                       # the primary's thread count, every initial weight one ulp off
        others.append(_solver_run({}, ("judge", "ssl"), ulp_seed=u))
        cold_others.append(_solver_run(EARLY, ("ssl",), ulp_seed=u))
    rec["threads"] = dict(primary=primary_threads, spread=threads, ulp_seeds=ulps,
                          spread_order=["%d threads" % t for t in threads] + ["weights one ulp off (seed %d)" % u for u in ulps])
    rec["spread"] = dict(sup=_spread(rec, others, "sup"), ssl_summaries=_spread(rec, others, "ssl_summaries"),
                         judge_val_loss=[[o["judge"][i]["val_loss"] for o in others] for i in range(len(rec["judge"]))],
                         judge_train_loss=[[float(np.mean(o["judge"][i]["losses"])) for o in others] for i in range(len(rec["judge"]))],
                         sup_best_cer=[o["sup_best_cer"] for o in others],
                         sup_final_norms=[o["sup_final_norms"] for o in others],
                         judge_final_norms=[o["judge_final_norms"] for o in others],
                         ssl_final_norms=[o["ssl_final_norms"] for o in others])
    rec["ssl_early"] = dict(sup=cold["sup"], ssl_steps=cold["ssl_steps"], ssl_summaries=cold["ssl_summaries"], ssl_final_norms=cold["ssl_final_norms"],
                           spread=dict(ssl_summaries=_spread(cold, cold_others, "ssl_summaries"),
                                       ssl_final_norms=[o["ssl_final_norms"] for o in cold_others],
                                       ssl_steps=[[o["ssl_steps"][i]["loss"] for o in cold_others]
                                                  for i in range(len(cold["ssl_steps"]))]))
    with open(os.path.join(HERE, "solver_run.json"), "w") as f:
        json.dump(rec, f, indent=0)
    print("solver_run: %d epochs, CER %s, best %.4f, ssl CER %s, test CER %.4f (%.0f s)"
          % (len(rec["sup"]), " ".join("%.3f" % e["cer"] for e in rec["sup"]), rec["sup_best_cer"],
             " ".join("%.4f" % e["cer"] for e in rec["ssl_summaries"]), rec["test_cer"], rec["seconds"]))
    for i, (e, sp) in enumerate(zip(rec["sup"], rec["spread"]["sup"])):
        print("  epoch %2d CER %.4f | other thread counts %s | same hypotheses %s of %d"
              % (i, e["cer"], " ".join("%.4f" % c for c in sp["cer"]), sp["same_hyps"], len(e["hyps"])))
    for i, (e, sp) in enumerate(zip(cold["ssl_summaries"], rec["ssl_early"]["spread"]["ssl_summaries"])):
        print("  early ssl summary %d CER %.4f | %s | same hypotheses %s" % (i, e["cer"], " ".join("%.4f" % c for c in sp["cer"]),
                                                                          sp["same_hyps"]))
    drop = _solver_run(synth.SOLVER_RUN_DROPOUT, ())
    keep = dict(sup=[{k: e[k] for k in ("epoch", "tf_rate", "train_loss", "val_loss", "cer")} for e in drop["sup"]],
                sup_best_cer=drop["sup_best_cer"])
    with open(os.path.join(HERE, "solver_run_dropout.json"), "w") as f:
        json.dump(keep, f, indent=0)
    print("solver_run_dropout: CER %s" % " ".join("%.3f" % e["cer"] for e in drop["sup"]))


def gen_solver_run_extend():
    """Append runs of the reference with its initial weights perturbed by 1e-6 RELATIVE (GOLDEN_SPREAD_REL = "1e-6:911,912") to
    the spread of an existing solver_run.json: the size of the differences a from-scratch implementation has in every
    operation (the parity gate of the hot path is 1e-3; its kernels are held to ~1e-6).  How well the reference agrees with
    itself under THAT perturbation is how well any implementation can be asked to agree with it."""
    path = os.path.join(HERE, "solver_run.json")
    with open(path) as f:
        rec = json.load(f)
    spec = os.environ.get("GOLDEN_SPREAD_REL", "1e-6:911,912")
    kind = "weights"
    if spec.startswith("grad:"):
        kind, spec = "grad", spec[5:]
    rel, seeds = float(spec.split(":")[0]), [int(v) for v in spec.split(":")[1].split(",")]
    EARLY = dict(epochs=1, ssl_iterations=30, summary_steps=10)
    early = rec["ssl_early"]
    for sd in seeds:
        kw = dict(rel=(rel, sd)) if kind == "weights" else dict(grad_noise=(rel, sd))
        o = _solver_run({}, ("judge", "ssl"), **kw)
        c = _solver_run(EARLY, ("ssl",), **kw)
        for key, primary, other, sp in (("sup", rec, o, rec["spread"]), ("ssl_summaries", rec, o, rec["spread"]),
                                        ("ssl_summaries", early, c, early["spread"])):
            add = _spread(primary, [other], key)
            for dst, a in zip(sp[key], add):
                for fld in ("cer", "val_loss", "train_loss", "same_hyps"):
                    dst[fld] += a[fld]
        for i in range(len(rec["judge"])):
            rec["spread"]["judge_val_loss"][i].append(o["judge"][i]["val_loss"])
            rec["spread"]["judge_train_loss"][i].append(float(np.mean(o["judge"][i]["losses"])))
        rec["spread"]["sup_best_cer"].append(o["sup_best_cer"])
        for k in ("sup_final_norms", "judge_final_norms", "ssl_final_norms"):
            rec["spread"][k].append(o[k])
        early["spread"]["ssl_final_norms"].append(c["ssl_final_norms"])
        for i in range(len(early["ssl_steps"])):
            early["spread"]["ssl_steps"][i].append(c["ssl_steps"][i]["loss"])
        rec["threads"]["spread_order"].append(("weights x (1 + %g N(0,1)) (seed %d)" if kind == "weights" else
                                               "every gradient + %g max|g| N(0,1), every step (seed %d)") % (rel, sd))
        rec["threads"].setdefault("rel_runs" if kind == "weights" else "grad_noise_runs", []).append(len(rec["threads"]["spread_order"]) - 1)
    with open(path, "w") as f:
        json.dump(rec, f, indent=0)
    for i, (e, sp) in enumerate(zip(rec["sup"], rec["spread"]["sup"])):
        print("  epoch %2d CER %.4f | %s | same hypotheses %s" % (i, e["cer"], " ".join("%.4f" % c for c in sp["cer"]), sp["same_hyps"]))
    for i, (e, sp) in enumerate(zip(early["ssl_summaries"], early["spread"]["ssl_summaries"])):
        print("  early ssl summary %d CER %.4f | %s | same hypotheses %s" % (i, e["cer"], " ".join("%.4f" % c for c in sp["cer"]), sp["same_hyps"]))


def gen_solver_loops():
    """The CONTROL FLOW of the reference's loops with the compute scripted (a stub per step / validation that returns given
    numbers): which teacher-forcing rate each epoch gets (solver.py:414-418), when the best model is saved and under which
    names (446-456: strict <, from 200), the judge's learning rate per epoch (MultiStepLR stepped at the epoch's start,
    308-315) and its saves (347-357: from 100), the semi-supervised loop's learning rate, summary steps and saves
    (519-563: from 2).  Deterministic, so the product's loops are held to it exactly (tests/test_solver_loops_cpu.py)."""
    import contextlib
    import io
    import tempfile
    import warnings
    import yaml
    import solver as ref_solver
    script = synth.SOLVER_LOOPS
    with open("/root/reference/config.yaml") as f:
        base = yaml.safe_load(f)
    out = {}
    with tempfile.TemporaryDirectory() as root:
        synth.write_solver_run_corpus(root, sizes=script["corpus"])
        cfg = synth.solver_run_config(base, root, **script["config"])
        with contextlib.redirect_stdout(io.StringIO()):
            warnings.simplefilter("ignore")
            s = ref_solver.Solver(cfg)
            log = []
            s.save_model = lambda path: log.append(["save_model", os.path.relpath(path, root)])
            s.save_judge = lambda path: log.append(["save_judge", os.path.relpath(path, root)])
            vals = iter(script["sup_cers"])
            s.sup_train_one_epoch = lambda e, tf: (log.append(["epoch", int(e), float(tf)]), 1.0)[1]
            s.validation = lambda: (log.append(["validation"]), (0.5, next(vals), ["a"], ["a"]))[1]
            _, best = s.sup_pretrain()
            out["sup"] = dict(log=list(log), best_cer=float(best))
            del log[:]
            jvals = iter(script["judge_val_losses"])
            s.judge_train_one_iteration = lambda ys: (log.append(["step", float(s.dis_opt.param_groups[0]["lr"])]),
                                                      dict(loss=1.0, avg_prob=0.1))[1]
            s.lm_validation = lambda: (log.append(["lm_validation"]), (next(jvals), ["a"]))[1]
            s.judge_pretrain()
            out["judge"] = dict(log=list(log), steps_per_epoch=len(s.train_unlab_y_loader))
            del log[:]
            svals = iter(script["ssl_cers"])
            s.validation = lambda: (log.append(["validation"]), (0.5, next(svals), ["a"], ["a"]))[1]
            s.ssl_train_one_iteration = lambda iteration: (log.append(["iteration", int(iteration),
                                                                        float(s.gen_opt.param_groups[0]["lr"])]),
                                                           dict(sup_loss=1.0, unsup_loss=0.5, loss=1.5))[1]
            s.ssl_train()
            out["ssl"] = dict(log=list(log))
    with open(os.path.join(HERE, "solver_loops.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("solver_loops: sup %d events, judge %d, ssl %d" % (len(out["sup"]["log"]), len(out["judge"]["log"]),
                                                              len(out["ssl"]["log"])))


def gen_text():
    """utils.py:192-235 helpers on a toy vocabulary."""
    vocab = {"<PAD>": 0, "<BOS>": 1, "<EOS>": 2, "a": 3, "b": 4, "c": 5, "<space>": 6, "<NOISE>": 7,
             "'": 8}
    nls = ["<NOISE>", "<PAD>", "<BOS>", "<EOS>"]
    preds = [[3, 4, 6, 5, 2, 3, 3], [7, 3, 3, 8, 4], [2, 3, 4], [5, 6, 6, 3, 2, 2]]
    refs = [[3, 4, 6, 5], [3, 8, 4, 4], [3], [5, 6, 3, 4]]
    cut = ref_utils.remove_pad_eos(preds, eos=2)
    hyp = ref_utils.to_sents(cut, vocab, nls)
    ref = ref_utils.to_sents(refs, vocab, nls)
    cer = ref_utils.calculate_cer(hyp, ref)
    mask = ref_utils._seq_mask([3, 1, 4], 5).numpy().tolist()
    with open(os.path.join(HERE, "text.json"), "w") as f:
        json.dump(dict(vocab=vocab, non_lang_syms=nls, preds=preds, refs=refs, cut=cut, hyp=hyp,
                       ref=ref, cer=cer, seq_mask=mask), f, indent=1)
    print("text: cer %.6f" % cer)


if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", "4")))
    jobs = dict(tiny_e2e=gen_tiny_e2e, tiny_lm=gen_tiny_lm, tiny_ssl=gen_tiny_ssl, cfg1=gen_cfg1, text=gen_text,
                tiny_opt=gen_tiny_opt, cfg2=lambda: gen_big("cfg2", synth.CFG2_SHAPE),
                cfg5=lambda: gen_big("cfg5", synth.CFG5_SHAPE), big_ssl=gen_big_ssl,
                solver_run=gen_solver_run, solver_loops=gen_solver_loops,
                solver_run_extend=gen_solver_run_extend)
    for name in (sys.argv[1:] or [j for j in jobs if j != "solver_run_extend"]):          # no arguments: everything (cfg2 / cfg5 take minutes)
        jobs[name]()
