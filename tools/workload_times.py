"""Step time of the hot path on the other BASELINE.json configurations (one GPU): cfg-1 (config.yaml defaults: 1x128
encoder, dec 320, batch 4, T=200), cfg-2 (the bench line), cfg-5 (T=1600, batch 8) and the semi-supervised generator
step of cfg-4 at cfg-2's shape (labeled 32x800 + unlabeled 32x800, judge LM 2x640).  One line per workload.

    python tools/workload_times.py [cfg1 cfg2 cfg5 ssl decode judge]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semi-supervised-asr_amd"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np
import torch
import bench
import __graft_entry__ as entry

entry.build()
import model as M
import parallel
import synth
from parallel import FlatAdam

dev = torch.device("cuda", 0)
CFG1 = dict(bench.CFG2, enc_hidden_dim=128, enc_n_layers=1, subsample=[2], dec_hidden_dim=320, att_dim=320, att_odim=320)
WORK = {"cfg1": (CFG1, 4, 200), "cfg2": (bench.CFG2, 32, 800), "cfg5": (bench.CFG2, 8, 1600), "ssl": (bench.CFG2, 32, 800),
        "decode": (bench.CFG2, 32, 800)}


def timed(step, warm=3, n=8):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


def run_judge():
    """judge_train_one_iteration (solver.py:288-301): the 2 x 640 LM on a text batch of 32 transcripts (cfg-2's label
    lengths), loss, backward, clip + Adam."""
    import hip_backend as hb
    cfg = bench.CFG2
    judge = M.LM(output_dim=cfg["output_dim"], embedding_dim=256, hidden_dim=640, dropout_rate=0.5, n_layers=2, bos=1,
                 eos=2, pad=0, ls_weight=0.05, labeldist=synth.labeldist(cfg["output_dim"], 6)).to(dev).train()
    opt = FlatAdam(judge, lr=1e-3, weight_decay=1e-6, amsgrad=True, max_grad_norm=5.0)
    _, _, ys = bench.global_batch(32, 800, 77)
    ys_d = [torch.from_numpy(y).to(dev) for y in ys]

    def step():
        _, lp, _ = judge(ys=ys_d, discrete_input=True)
        loss = -judge.mask_and_cal_sum(lp, ys=ys_d, mask=None)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss
    hb.LAUNCHES.clear()
    ms, loss = timed(step)
    print("judge B=32 L=%d: %.2f ms/step (loss %.4f)  paths %s" % (max(len(y) for y in ys) + 5, ms, float(loss.detach()),
          {k: v for k, v in hb.LAUNCHES.items() if "lstm" in k}), flush=True)


def run(name):
    if name == "judge":
        return run_judge()
    cfg, B, T = WORK[name]
    torch.manual_seed(1000)
    net = M.E2E(labeldist=synth.labeldist(cfg["output_dim"], 5), **cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.e2e_weights(cfg, 99).items()})
    net = net.to(dev).train()
    if name == "ssl":
        net.decoder.output_layer.bias.data[2] = -10.0     # random weights: keep the greedy hypothesis off <EOS> (mask sum > 0)
    # ssl: random weights + random labels drift to an all-<EOS> hypothesis within a few steps (mask sum 0 -> the
    # reference's 0/0); a negligible learning rate keeps the timed iterations on the same regime
    opt = FlatAdam(net, lr=5e-4 if name != "ssl" else 1e-8, weight_decay=1e-6, amsgrad=True, max_grad_norm=5.0)
    xs, lens, ys = bench.global_batch(B, T, 1234)
    xs_r, lens_r, ys_r, info = parallel.shard_batch(xs, lens, ys, 0, 1)
    xs_d = torch.from_numpy(np.ascontiguousarray(xs_r)).to(dev)
    ys_d = [torch.from_numpy(y).to(dev) for y in ys_r]
    tl = M.padded_lengths(info["t_max"], cfg["enc_n_layers"], cfg["subsample"])
    if name == "decode":
        # validation decode (solver.py:212-242): eval mode, no autograd, greedy, a fixed 230 steps like the reference
        net.eval()

        def dec():
            with torch.no_grad():
                return net(xs_d, lens_r, ys=None, max_dec_timesteps=230)[2]
        ms, pred = timed(dec)
        print("%-6s B=%d T=%d greedy x 230 steps: %.2f ms/batch = %.0f utt/s" % (name, B, T, ms, B / ms * 1e3), flush=True)
        # a trained model ends its hypotheses and a group of 4 utterances then stops (hip_backend.DECODE_EARLY_STOP);
        # random weights never emit <EOS>, so the lower bound is shown by forcing <EOS> with the output bias
        net.decoder.output_layer.bias.data[2] = 50.0
        ms0, _ = timed(dec)
        print("%-6s ... every utterance at <EOS> after the first step: %.2f ms/batch (encoder + 1 decoder step)" % (name, ms0),
              flush=True)
        return
    if name != "ssl":
        def step():
            _, lp, _, _ = net(xs_d, lens_r, ys_d, tf_rate=1.0, total_length=tl, olength=info["olength"])
            loss = parallel.local_loss(lp, info)
            opt.zero_grad()
            loss.backward()
            opt.step()
            return loss
        ms, loss = timed(step)
        print("%-5s B=%d T=%d L+1=%d: %.2f ms/step = %.0f utt/s  (loss %.4f)" % (name, B, T, info["olength"], ms,
              B / ms * 1e3, float(loss.detach())), flush=True)
        return
    # cfg-4: solver.gen_train_one_iteration on a second (unlabeled) batch of the same shape, judge as config.yaml
    judge = M.LM(output_dim=cfg["output_dim"], embedding_dim=256, hidden_dim=640, dropout_rate=0.5, n_layers=2, bos=1,
                 eos=2, pad=0, ls_weight=0.05, labeldist=synth.labeldist(cfg["output_dim"], 6)).to(dev).train()
    uxs, ulens, _ = bench.global_batch(B, T, 4321)
    uxs_d = torch.from_numpy(uxs).to(dev)
    proportion = 0.125
    parts = {}

    def step():
        e = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        e[0].record()
        _, u_lp, u_pred, _ = net(uxs_d, ulens, ys=None, sample=False, label_smoothing=False,
                                 max_dec_timesteps=int(uxs_d.size(1) * proportion), smooth=True, scaling=3)
        e[1].record()
        with torch.no_grad():                       # as solver.gen_train_one_iteration: the judge only scores
            _, lm_probs, _ = judge(ys=u_pred, discrete_input=False)
        mask = (u_pred != 2).float()
        unsup = -torch.sum(lm_probs * u_lp * mask) / torch.sum(mask)
        e[2].record()
        _, lab_lp, _, _ = net(xs_d, lens_r, ys=ys_d, tf_rate=1.0, sample=False)
        loss = -torch.mean(lab_lp) + 0.001 * unsup
        e[3].record()
        opt.zero_grad()
        loss.backward()
        opt.step()
        e[4].record()
        parts["e"] = e
        return loss
    ms, loss = timed(step)
    e = parts["e"]
    names = ["unlabeled smooth-greedy forward", "judge forward", "labeled forward", "backward + step"]
    det = ", ".join("%s %.2f" % (n, e[i].elapsed_time(e[i + 1])) for i, n in enumerate(names))
    print("%-5s B=%d+%d T=%d: %.2f ms/iteration = %.0f utt/s (labeled+unlabeled)  (loss %.4f)\n      %s" % (
        name, B, B, T, ms, 2 * B / ms * 1e3, float(loss.detach()), det), flush=True)


for w in (sys.argv[1:] or ["cfg1", "cfg2", "cfg5", "ssl", "decode", "judge"]):
    run(w)
