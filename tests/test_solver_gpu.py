"""End-to-end run of the reference-surface Solver on synthetic pickles (GPU): supervised epoch + validation CER,
judge pre-training epoch, semi-supervised iterations, test(), checkpoint round trip."""
import os
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _vocab():
    syms = ["<PAD>", "<BOS>", "<EOS>"] + [chr(ord("a") + i) for i in range(8)] + ["<space>", "<NOISE>"]
    return {s: i for i, s in enumerate(syms)}


def _write_data(root, vocab):
    from dataset import synthetic_utterances
    for name, n, seed in (("train", 12, 1), ("dev", 6, 2), ("eval", 3, 3)):
        data = synthetic_utterances(n, 16, len(vocab), 40, seed)
        with open(os.path.join(root, name + ".pkl"), "wb") as f:
            pickle.dump(data, f)
    with open(os.path.join(root, "vocab_dict.pkl"), "wb") as f:
        pickle.dump(vocab, f)
    with open(os.path.join(root, "non_lang_syms.pkl"), "wb") as f:
        pickle.dump(["<NOISE>", "<PAD>", "<BOS>", "<EOS>"], f)


def _config(root):
    import yaml
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(here, "semi-supervised-asr_amd", "config.yaml")) as f:
        cfg = yaml.safe_load(f)
    cfg.update(logdir=os.path.join(root, "log"), model_dir=root, model_name="m", load_model_path=os.path.join(root, "m"),
               load_judge_path=os.path.join(root, "m"), dataset_root_dir=root,
               vocab_path=os.path.join(root, "vocab_dict.pkl"),
               non_lang_syms_path=os.path.join(root, "non_lang_syms.pkl"), labeled_set="train",
               unlabeled_speech_set="train", unlabeled_text_set="train", dev_set="dev", test_set="eval",
               min_feature_length=4, max_dec_timesteps=8, batch_size=4, input_dim=16, enc_hidden_dim=16,
               enc_n_layers=2, subsample=[2, 2], dec_hidden_dim=16, att_dim=16, att_odim=16, conv_channels=2,
               conv_kernel_size=3, embedding_dim=16, dis_hidden_dim=16, dis_embedding_dim=16, epochs=1, judge_epochs=1,
               ssl_iterations=2, summary_steps=2, lm_sample_steps=5)
    return cfg


def test_solver_end_to_end(tmp_path, monkeypatch):
    import __graft_entry__ as entry
    entry.build()
    assert torch.cuda.is_available()
    from solver import Solver
    root = str(tmp_path)
    vocab = _vocab()
    _write_data(root, vocab)
    monkeypatch.chdir(root)
    torch.manual_seed(0)
    np.random.seed(0)
    solver = Solver(_config(root))
    assert abs(solver.labeldist.sum() - 1.0) < 1e-9 and solver.labeldist[vocab["<PAD>"]] == 0
    w_before = solver.model.decoder.output_layer.weight.detach().clone()
    best, cer = solver.sup_pretrain()
    assert np.isfinite(cer) and best is not None
    assert not torch.equal(w_before, solver.model.decoder.output_layer.weight.detach())
    assert os.path.exists(os.path.join(root, "m.ckpt")) and os.path.exists(os.path.join(root, "m.opt"))
    solver.judge_pretrain()
    assert os.path.exists(os.path.join(root, "m.judge.ckpt"))
    solver.ssl_train()
    cer = solver.test(state_dict=solver.model.state_dict())
    assert np.isfinite(cer) and os.path.exists(os.path.join(root, "eval.txt"))
    # checkpoint round trip incl. optimiser state in torch.optim.Adam's schema
    solver2 = Solver(_config(root), load_model=True)
    for (n1, p1), (n2, p2) in zip(solver.model.state_dict().items(), solver2.model.state_dict().items()):
        assert n1 == n2
    sd = solver.gen_opt.state_dict()
    assert set(sd["state"][0].keys()) >= {"step", "exp_avg", "exp_avg_sq", "max_exp_avg_sq"}


@pytest.mark.parametrize("poison_loss", [True, False])
def test_solver_recovers_from_aborted_persistent_kernel(tmp_path, monkeypatch, poison_loss):
    """A persistent kernel that aborts poisons its output with NaN and sets the STICKY abort latch (csrc/persist.h: the
    per-launch abort word is zeroed before the next launch, the latch is not); the solver reads loss and latch together
    before the optimiser step, switches this process to the per-step kernels and repeats the step (the parameters never
    see the NaN).  poison_loss=False is the case ADVICE r2 describes: the abort happened in a launch whose NaN never
    reached the loss (e.g. a backward kernel) - the latch alone must trigger the repeat."""
    import __graft_entry__ as entry
    entry.build()
    import hip_backend as hb
    from solver import Solver
    root = str(tmp_path)
    vocab = _vocab()
    _write_data(root, vocab)
    monkeypatch.chdir(root)
    torch.manual_seed(0)
    np.random.seed(0)
    solver = Solver(_config(root))
    dev = next(solver.model.parameters()).device
    state = {"calls": 0}
    real_forward = solver._sharded_forward

    def aborted_once(xs, ilens, ys, tf_rate):
        loss = real_forward(xs, ilens, ys, tf_rate)
        state["calls"] += 1
        if state["calls"] == 2:
            latch = hb.persist_scratch(dev)[1]
            latch[0], latch[1] = 1, 7                    # what raise_abort() leaves behind
            return loss * float("nan") if poison_loss else loss
        return loss

    flags = (hb.USE_PERSIST, hb.USE_PERSIST_DEC, hb.USE_PERSIST_DEC_BWD)
    monkeypatch.setattr(solver, "_sharded_forward", aborted_once)
    try:
        hb.persist_clear_abort(dev)
        mean_loss = solver.sup_train_one_epoch(0, 1.0)
        steps = len(solver.train_lab_loader)
        # the host reads a step's record while the next step runs (Solver._step): the abort of step 2 is found after
        # step 3 was enqueued; neither was applied (the Adam kernel checks the sticky latch on the device), both are repeated
        assert state["calls"] == steps + 2, "the aborted step and the one enqueued behind it are repeated once"
        assert np.isfinite(mean_loss)
        assert not (hb.USE_PERSIST or hb.USE_PERSIST_DEC or hb.USE_PERSIST_DEC_BWD)
        assert not hb.persist_aborted(dev), "the latch is cleared once the abort has been dealt with"
        for name, prm in solver.model.named_parameters():
            assert torch.isfinite(prm).all(), name
    finally:
        hb.USE_PERSIST, hb.USE_PERSIST_DEC, hb.USE_PERSIST_DEC_BWD = flags


def test_greedy_decode_repeats_after_an_abort_in_an_earlier_launch(tmp_path, monkeypatch):
    """Greedy decoding is several persistent launches (three encoder layers, then the decoder) and reads no loss.  An
    abort in any of them but the last used to be invisible (every launch zeroes the per-launch abort word); with the
    sticky latch Solver._greedy sees it after the decode, falls back to the per-step kernels and decodes again."""
    import __graft_entry__ as entry
    entry.build()
    import hip_backend as hb
    from solver import Solver
    from utils import to_gpu
    root = str(tmp_path)
    _write_data(root, _vocab())
    monkeypatch.chdir(root)
    torch.manual_seed(0)
    np.random.seed(0)
    solver = Solver(_config(root))
    xs, ilens, _ = to_gpu(next(iter(solver.dev_loader)))
    dev = xs.device
    flags = (hb.USE_PERSIST, hb.USE_PERSIST_DEC, hb.USE_PERSIST_DEC_BWD)
    try:
        hb.persist_clear_abort(dev)
        solver.model.eval()
        want = solver._greedy(xs, ilens)
        assert (hb.USE_PERSIST, hb.USE_PERSIST_DEC, hb.USE_PERSIST_DEC_BWD) == flags, "no abort: no fallback"
        real_encoder = solver.model.encoder.forward

        def encoder_then_abort(*a, **k):
            out = real_encoder(*a, **k)
            latch = hb.persist_scratch(dev)[1]
            latch[0], latch[1] = 1, 1                    # an encoder layer aborted; the decoder launch follows
            return out

        monkeypatch.setattr(solver.model.encoder, "forward", encoder_then_abort)
        got = solver._greedy(xs, ilens)
        assert not (hb.USE_PERSIST or hb.USE_PERSIST_DEC), "the decode was repeated on the per-step kernels"
        assert got == want
    finally:
        hb.USE_PERSIST, hb.USE_PERSIST_DEC, hb.USE_PERSIST_DEC_BWD = flags
        hb.persist_clear_abort(dev)


# ------------------------------------------------------------------------------------------------------------------
# The Solver's own step methods against the reference's numbers (tests/golden/tiny_ssl.npz, tiny_lm.npz) and against
# themselves under data parallelism.
def _tiny_solver(root, monkeypatch, t=None, l=None, seeds=(11, 31, 12, 32), **over):
    """Solver whose model / judge have the shapes of synth.TINY / synth.TINY_LM (or the given dims) and the fixtures'
    weights.  seeds = (model weights, judge weights, model labeldist, judge labeldist)."""
    import synth
    from solver import Solver
    t = synth.TINY if t is None else t
    l = synth.TINY_LM if l is None else l
    nv = t["output_dim"]
    syms = ["<PAD>", "<BOS>", "<EOS>"] + ["s%d" % i for i in range(nv - 5)] + ["<space>", "<NOISE>"]
    vocab = {s: i for i, s in enumerate(syms)}
    assert len(vocab) == nv
    from dataset import synthetic_utterances
    for name, n, seed in (("train", 12, 1), ("dev", 4, 2)):
        with open(os.path.join(root, name + ".pkl"), "wb") as f:
            pickle.dump(synthetic_utterances(n, t["input_dim"], len(vocab), 24, seed), f)
    with open(os.path.join(root, "vocab_dict.pkl"), "wb") as f:
        pickle.dump(vocab, f)
    with open(os.path.join(root, "non_lang_syms.pkl"), "wb") as f:
        pickle.dump(["<NOISE>", "<PAD>", "<BOS>", "<EOS>"], f)
    cfg = _config(root)
    cfg.update(input_dim=t["input_dim"], enc_hidden_dim=t["enc_hidden_dim"], enc_n_layers=t["enc_n_layers"],
               subsample=t["subsample"], dropout_rate=0.0, dec_hidden_dim=t["dec_hidden_dim"], att_dim=t["att_dim"],
               conv_channels=t["conv_channels"], conv_kernel_size=t["conv_kernel_size"], att_odim=t["att_odim"],
               embedding_dim=t["embedding_dim"], ls_weight=t["ls_weight"], dis_embedding_dim=l["embedding_dim"],
               dis_hidden_dim=l["hidden_dim"], dis_dropout_rate=0.0, dis_layers=l["n_layers"], d_learning_rate=2e-4,
               learning_rate=5e-4, weight_decay=1e-6, max_grad_norm=5, unsup_weight=0.5, smooth_embedding=True,
               softmax_scaling=3, min_feature_length=1, add_gaussian=False)
    cfg.update(over)
    monkeypatch.chdir(root)
    solver = Solver(cfg)
    dev = next(solver.model.parameters()).device
    with torch.no_grad():
        wm, wj = synth.e2e_weights(t, seeds[0]), synth.lm_weights(l, seeds[1])
        for k, v in solver.model.state_dict().items():
            v.copy_(torch.from_numpy(wm[k]))
        for k, v in solver.judge.state_dict().items():
            v.copy_(torch.from_numpy(wj[k]))
    for mod, seed in ((solver.model.decoder, seeds[2]), (solver.judge, seeds[3])):
        mod.labeldist = synth.labeldist(nv, seed)
        mod.vlabeldist = torch.from_numpy(np.asarray(mod.labeldist, dtype=np.float32)).to(dev)
    solver.model.decoder._dist_dev = {}
    solver.judge._dist_dev = {}
    solver.proportion = 0.5
    return solver, dev


def _golden(name):
    here = os.path.dirname(os.path.abspath(__file__))
    return dict(np.load(os.path.join(here, "golden", name), allow_pickle=False))


def test_gen_train_one_iteration_against_golden(tmp_path, monkeypatch):
    """Solver.gen_train_one_iteration itself (reference solver.py:460-495): its three losses and the
    gradient it hands to the optimiser, against what the reference computed on the same two batches."""
    import __graft_entry__ as entry
    entry.build()
    import synth
    g = _golden("tiny_ssl.npz")
    solver, dev = _tiny_solver(str(tmp_path), monkeypatch)
    xs, ilens, ys = synth.batch(8, 9, synth.TINY_ILENS, synth.TINY_YLENS, 13)
    uxs, uilens, _ = synth.batch(8, 9, [12, 10, 7], [2, 2, 2], 41)
    before = {n: p.detach().clone() for n, p in solver.model.named_parameters()}
    judge_before = {n: p.detach().clone() for n, p in solver.judge.named_parameters()}
    np.random.seed(9)
    meta = solver.gen_train_one_iteration(torch.from_numpy(xs).to(dev), ilens, [torch.from_numpy(y).to(dev) for y in ys],
                                          torch.from_numpy(uxs).to(dev), uilens)
    assert abs(meta["sup_loss"] - float(g["sup"])) <= 1e-5 * abs(float(g["sup"]))
    assert abs(meta["unsup_loss"] - float(g["unsup"])) <= 1e-4 * abs(float(g["unsup"]))
    assert abs(meta["loss"] - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    for n, p in solver.model.named_parameters():       # the flat gradient buffer still holds what the step consumed
        want = g["grad/" + n]
        err = float((p.grad.cpu() - torch.from_numpy(want)).abs().max())
        assert err <= 1e-3 * float(np.abs(want).max()) + 1e-6, (n, err)
        assert not torch.equal(p.detach(), before[n]) or float(np.abs(want).max()) == 0.0, n      # generator stepped
    for n, p in solver.judge.named_parameters():       # ... and the judge is not (solver.py:486-489)
        assert torch.equal(p.detach(), judge_before[n]), n


def test_gen_train_one_iteration_at_cfg4_width_against_golden(tmp_path, monkeypatch):
    """The semi-supervised generator step at the width cfg-4 runs (3x512 model + 2x640 judge, config.yaml:26-34) through
    Solver.gen_train_one_iteration, against what the reference's arithmetic produced at that width
    (tests/golden/big_ssl.npz, make_golden.py gen_big_ssl; solver.py:460-495): the three losses and every generator
    gradient (norm, ends, seeded 4 096-element sample).  The launch counters must show the kernels this workload is
    meant to run on: the free-running persistent decoder (forward AND backward, the smooth-embedding feedback carried
    inside the kernels), the persistent teacher-forced decoder, the persistent encoder recurrences of both passes, the
    judge's H = 640 persistent forward."""
    import __graft_entry__ as entry
    entry.build()
    import synth
    import hip_backend as hb
    g = _golden("big_ssl.npz")
    sh = synth.BIG_SSL_SHAPE
    solver, dev = _tiny_solver(str(tmp_path), monkeypatch, t=synth.CFG2, l=synth.CFG_JUDGE,
                               seeds=(sh["wseed"], sh["jseed"], sh["ldseed"], sh["jldseed"]),
                               unsup_weight=sh["unsup_weight"], softmax_scaling=sh["scaling"])
    solver.proportion = sh["proportion"]
    xs, ilens, ys = synth.ragged_batch(sh["n_lab"], sh["t_max"], 80, 34, sh["bseed"])
    uxs, uilens, _ = synth.ragged_batch(sh["n_unlab"], sh["t_max"], 80, 34, sh["ubseed"])
    judge_before = {n: p.detach().clone() for n, p in solver.judge.named_parameters()}
    hb.persist_clear_abort(dev)
    hb.LAUNCHES.clear()
    np.random.seed(9)
    with hb.require_persistent():
        meta = solver.gen_train_one_iteration(torch.from_numpy(xs).to(dev), ilens, [torch.from_numpy(y).to(dev) for y in ys],
                                              torch.from_numpy(uxs).to(dev), uilens)
    assert not hb.persist_aborted(dev)
    launches = dict(hb.LAUNCHES)
    assert launches.get("dec_free_persist") == 1 and launches.get("dec_fwd_persist") == 1, launches
    assert launches.get("dec_bwd_persist") == 2 and "dec_bwd_step" not in launches, launches
    assert launches.get("lstm_fwd_persist") == 3 + 3 + 2 and launches.get("lstm_bwd_persist") == 3 + 3, launches
    assert abs(meta["sup_loss"] - float(g["sup"])) <= 1e-5 * abs(float(g["sup"])), (meta, float(g["sup"]))
    assert abs(meta["unsup_loss"] - float(g["unsup"])) <= 1e-4 * abs(float(g["unsup"])), (meta, float(g["unsup"]))
    assert abs(meta["loss"] - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    worst = 0.0
    for i, (n, p) in enumerate(solver.model.named_parameters()):      # the flat gradient buffer still holds what the step consumed
        flat = p.grad.detach().cpu().numpy().ravel()
        norm = float(np.sqrt((flat.astype(np.float64) ** 2).sum()))
        assert abs(norm - float(g["gnorm/" + n])) <= 1e-3 * float(g["gnorm/" + n]), (n, norm, float(g["gnorm/" + n]))
        scale = float(g["gmax/" + n])
        e = max(np.abs(flat[:16] - g["ghead/" + n]).max(), np.abs(flat[-16:] - g["gtail/" + n]).max(),
                np.abs(flat[synth.grad_sample_index(i, flat.size)] - g["gsample/" + n]).max()) / scale
        worst = max(worst, float(e))
        assert e <= 1e-3, (n, float(e))
    for n, p in solver.judge.named_parameters():       # the judge is not stepped (solver.py:486-489)
        assert torch.equal(p.detach(), judge_before[n]), n
    print("big_ssl: %s, worst gradient element error %.2e, launches %s" % (meta, worst, launches))


def test_judge_train_one_iteration_against_golden(tmp_path, monkeypatch):
    """Solver.judge_train_one_iteration (solver.py:288-301) vs the reference: loss, average probability,
    weights after the step."""
    import __graft_entry__ as entry
    entry.build()
    g = _golden("tiny_lm.npz")
    solver, dev = _tiny_solver(str(tmp_path), monkeypatch)
    ys = [torch.from_numpy(g["ys%d" % i]).to(dev) for i in range(3)]
    meta = solver.judge_train_one_iteration(ys)
    assert abs(meta["loss"] - float(g["d_loss"])) <= 1e-5 * abs(float(g["d_loss"]))
    assert abs(meta["avg_prob"] - float(g["d_avg_prob"])) <= 1e-5 * abs(float(g["d_avg_prob"]))
    for n, p in solver.judge.named_parameters():
        want = g["after1/" + n]
        assert float((p.detach().cpu() - torch.from_numpy(want)).abs().max()) <= 1e-4 * float(np.abs(want).max()) + 2e-6, n


def _dp_worker(rank, world, port, root, out, use_feed=False):
    """One rank of a 2-rank rehearsal that SHARES the one GPU (gloo for the collectives; init_distributed switches a
    shared card to the per-step kernels): supervised step with scheduled sampling (tf_rate 0.5, numpy RNG seeded by the
    Solver), judge step, semi-supervised step -> rank 0 saves the resulting weights and the reported scalars.
    use_feed: the batches come from the Solver's input pipeline (feed.DeviceFeed: every rank pads and uploads ITS rows
    only, parallel.LocalShard) instead of the global batch uploaded by every rank."""
    import sys
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (here, os.path.join(here, "semi-supervised-asr_amd"), os.path.join(here, "tests", "golden"),
              os.path.join(here, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank), ASR_DIST_BACKEND="gloo")
    import pytest as _pytest
    mpatch = _pytest.MonkeyPatch()
    try:
        torch.manual_seed(0)
        # per-GPU batch 3 on 2 ranks == batch 6 in one process; numpy_seed makes the single process draw like the ranks
        # dp_overlap: the 2-rank run exchanges its gradients in buckets issued from inside the backward pass
        solver, dev = _tiny_solver(root, mpatch, batch_size=6 // world, numpy_seed=0, shuffle=False, dp_overlap=True)
        assert solver.gen_opt.buf.overlap == (world > 1)
        scalars = []
        from utils import to_gpu, cc
        if use_feed:
            import parallel
            xs, ilens, ys = next(iter(solver._feed(solver.train_lab_loader)))
            texts = next(iter(solver._feed(solver.train_unlab_y_loader, kind="text")))
            texts = texts.xs if world > 1 else texts.ys
            uxs, uilens = next(iter(solver._feed(solver.train_unlab_x_loader, kind="speech")))
            if world > 1:
                assert isinstance(xs, parallel.LocalShard) and xs.xs.shape[0] == 3 and xs.info["b_global"] == 6
                assert isinstance(uxs, parallel.LocalShard) and isinstance(texts, parallel.LocalShard)
        else:
            xs, ilens, ys = to_gpu(next(iter(solver.train_lab_loader)))
            assert len(ilens) == 6
            texts = [cc(y) for y in next(iter(solver.train_unlab_y_loader))]
            uxs, uilens = next(iter(solver.train_unlab_x_loader))
            uxs = cc(uxs)

        def make_local():
            loss = solver._sharded_forward(xs, ilens, ys, 0.5)
            return loss, [loss]
        scalars += solver._step(make_local, solver.gen_opt, 1)
        meta = solver.judge_train_one_iteration(texts)
        scalars += [meta["loss"], meta["avg_prob"]]
        meta = solver.gen_train_one_iteration(xs, ilens, ys, uxs, uilens)
        scalars += [meta["unsup_loss"], meta["sup_loss"], meta["loss"]]
        solver.flush()        # every rank resolves its lazy scalars at the same point (a recovery is a sequence of collectives)
        if rank == 0:
            torch.save(dict(scalars=[float(v) for v in scalars], model={n: p.detach().cpu() for n, p in solver.model.named_parameters()},
                            judge={n: p.detach().cpu() for n, p in solver.judge.named_parameters()}), out)
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
    finally:
        mpatch.undo()


@pytest.mark.parametrize("use_feed", [False, True])
def test_solver_steps_two_ranks_equal_one_process(tmp_path, use_feed):
    """supervised (tf_rate 0.5 through the Solver's seeded numpy stream), judge and semi-supervised
    steps of the product Solver on 2 data-parallel ranks == the same steps in one process on the global batches.
    use_feed: the 2 ranks take rank-local batches from the input pipeline (each pads and uploads its 3 of
    the 6 utterances), the one process the global batch through the reference's to_gpu."""
    import socket
    import torch.multiprocessing as mp
    import __graft_entry__ as entry
    entry.build()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r2, r1 = str(tmp_path / "w2"), str(tmp_path / "w1")
    os.makedirs(r2), os.makedirs(r1)
    mp.spawn(_dp_worker, args=(2, port, r2, os.path.join(r2, "out.pt"), use_feed), nprocs=2, join=True)
    mp.spawn(_dp_worker, args=(1, 0, r1, os.path.join(r1, "out.pt")), nprocs=1, join=True)
    got, ref = torch.load(os.path.join(r2, "out.pt")), torch.load(os.path.join(r1, "out.pt"))
    for a, b in zip(got["scalars"], ref["scalars"]):
        assert abs(a - b) <= 2e-5 * max(1.0, abs(b)), (got["scalars"], ref["scalars"])
    for part in ("model", "judge"):
        for n, want in ref[part].items():
            err = float((got[part][n] - want).abs().max())
            assert err <= 1e-5 * float(want.abs().max()) + 2e-6, (part, n, err)


@pytest.mark.parametrize("thread", [True, False])
def test_epoch_through_the_input_pipeline_equals_the_to_gpu_loop(tmp_path, monkeypatch, thread):
    """Solver.sup_train_one_epoch takes its batches from feed.DeviceFeed (collated one step ahead into pinned
    memory, uploaded on a side stream, labels as one packed tensor) - same batches, same order, same numbers as the
    reference's loop `for data in loader: to_gpu(data)` (solver.py:365-367) around the same step; judge_pretrain's and
    validation's loops read through the same pipeline."""
    import __graft_entry__ as entry
    entry.build()
    from utils import to_gpu
    results = []
    for mode in ("feed", "to_gpu"):
        root = str(tmp_path / (mode + str(thread)))
        os.makedirs(root)
        torch.manual_seed(0)
        solver, dev = _tiny_solver(root, monkeypatch, batch_size=5, shuffle=True, numpy_seed=3, prefetch_thread=thread)
        if mode == "feed":
            mean = solver.sup_train_one_epoch(0, 0.7)
        else:
            losses = []
            for data in solver.train_lab_loader:
                xs, ilens, ys = to_gpu(data)
                losses.append(float(solver.sup_train_one_iteration(xs, ilens, ys, 0.7)))
            mean = sum(losses) / len(losses)
        solver.flush()
        val = solver.validation()
        results.append((mean, val[0], val[1], {n: p.detach().cpu().clone() for n, p in solver.model.named_parameters()}))
    (m1, v1, c1, w1), (m2, v2, c2, w2) = results
    # (not bitwise: the split-K products accumulate with atomics, run-to-run order differs)
    assert abs(m1 - m2) <= 1e-5 * abs(m2) and abs(v1 - v2) <= 1e-5 * abs(v2) and abs(c1 - c2) <= 0.05
    for n in w2:
        assert float((w1[n] - w2[n]).abs().max()) <= 1e-5 * float(w2[n].abs().max()) + 1e-6, n


def test_bucketed_pickle_epoch_on_the_gpu(tmp_path, monkeypatch):
    """SURVEY 8f-2 on the GPU: PickleDataset (dataset.py:8-80 of the reference) -> BucketBatchSampler (`bucket_batches`)
    -> feed.DeviceFeed -> Solver.sup_train_one_epoch.  The bucketed epoch sees every utterance exactly once, pads less than
    the reference's shuffled batches of the same pickle, and trains to the same numbers as the to_gpu loop over the same
    loader."""
    import __graft_entry__ as entry
    entry.build()
    from utils import to_gpu
    from dataloader import BucketBatchSampler, padded_fraction
    results = []
    for mode in ("feed", "to_gpu"):
        root = str(tmp_path / mode)
        os.makedirs(root)
        torch.manual_seed(0)
        solver, dev = _tiny_solver(root, monkeypatch, batch_size=4, shuffle=True, numpy_seed=5, bucket_batches=True)
        assert isinstance(solver.train_lab_loader.batch_sampler, BucketBatchSampler)
        if mode == "feed":
            seen = []
            keep = solver.sup_train_one_iteration

            def spy(xs, ilens, ys, *a, **k):
                seen.append([int(v) for v in ilens])
                return keep(xs, ilens, ys, *a, **k)
            monkeypatch.setattr(solver, "sup_train_one_iteration", spy)
            mean = solver.sup_train_one_epoch(0, 0.7)
            monkeypatch.setattr(solver, "sup_train_one_iteration", keep)
            lens = sorted(v for b in seen for v in b)
            want = sorted(int(solver.train_lab_dataset[i][0].shape[0]) for i in range(len(solver.train_lab_dataset)))
            assert lens == want, (lens, want)                                     # every utterance once
            assert all(b == sorted(b, reverse=True) for b in seen)                # collate order (dataloader.py:8-9)
            frames = sum(sum(b) for b in seen)
            bucketed = 1.0 - frames / float(sum(max(b) * len(b) for b in seen))
            n, every = len(want), [int(solver.train_lab_dataset[i][0].shape[0]) for i in range(len(want))]
            floor = padded_fraction(every, [list(range(i, min(i + 4, n))) for i in range(0, n, 4)])   # neighbours of the sorted keys
            strided = padded_fraction(every, [list(range(i, n, (n + 3) // 4)) for i in range((n + 3) // 4)])
            assert abs(bucketed - floor) <= 1e-9 and bucketed < strided, (bucketed, floor, strided)
        else:
            losses = []
            for data in solver.train_lab_loader:
                xs, ilens, ys = to_gpu(data)
                losses.append(float(solver.sup_train_one_iteration(xs, ilens, ys, 0.7)))
            mean = sum(losses) / len(losses)
        solver.flush()
        results.append((mean, {n: p.detach().cpu().clone() for n, p in solver.model.named_parameters()}))
    (m1, w1), (m2, w2) = results
    assert np.isfinite(m1) and abs(m1 - m2) <= 1e-5 * abs(m2), (m1, m2)
    for n in w2:
        assert float((w1[n] - w2[n]).abs().max()) <= 1e-5 * float(w2[n].abs().max()) + 1e-6, n


def test_solver_recovers_from_a_kernel_raised_abort(tmp_path, monkeypatch):
    """the abort of test_solver_recovers_from_aborted_persistent_kernel raised by the KERNEL - the armed
    fault of csrc/persist.h (ASR_DEBUG_FAULT in the launch's `arith`: a producer of the encoder's persistent LSTM forward goes
    silent, the bounded spins expire, raise_abort sets latch + code, the outputs are NaN) - and found by Solver._step one step late: both steps
    enqueued behind the abort were skipped on the device and are repeated on the per-step kernels."""
    import __graft_entry__ as entry
    entry.build()
    import hip_backend as hb
    from solver import Solver
    root = str(tmp_path)
    _write_data(root, _vocab())
    monkeypatch.chdir(root)
    torch.manual_seed(0)
    np.random.seed(0)
    cfg = _config(root)
    cfg.update(enc_hidden_dim=512, enc_n_layers=1, subsample=[2], dropout_rate=0.0)     # H = 512: the persistent LSTM kernels (and their FAULT instantiation)
    solver = Solver(cfg)
    dev = next(solver.model.parameters()).device
    state = {"calls": 0}
    real_forward = solver._sharded_forward

    def armed_once(xs, ilens, ys, tf_rate):
        state["calls"] += 1
        arm = state["calls"] == 2 and hb.USE_PERSIST
        old = hb.ARITH[0]
        if arm:
            hb.ARITH[0] = old | hb.DEBUG_FAULT
        try:
            return real_forward(xs, ilens, ys, tf_rate)
        finally:
            hb.ARITH[0] = old
    flags = (hb.USE_PERSIST, hb.USE_PERSIST_DEC, hb.USE_PERSIST_DEC_BWD)
    monkeypatch.setattr(solver, "_sharded_forward", armed_once)
    try:
        hb.persist_clear_abort(dev)
        hb.LAUNCHES.clear()
        mean_loss = solver.sup_train_one_epoch(0, 1.0)
        steps = len(solver.train_lab_loader)
        assert hb.LAUNCHES["lstm_fwd_persist"] >= 2, "the encoder ran on the persistent kernel before the fault"
        assert state["calls"] == steps + 2, "the aborted step and the one enqueued behind it are repeated once"
        assert np.isfinite(mean_loss)
        assert not (hb.USE_PERSIST or hb.USE_PERSIST_DEC or hb.USE_PERSIST_DEC_BWD)
        assert not hb.persist_aborted(dev)
        for name, prm in solver.model.named_parameters():
            assert torch.isfinite(prm).all(), name
        # the per-step kernels are a probation, not a verdict (hb.PERSIST_RETRY_STEPS): when it ends the next step runs on the
        # persistent kernels again - here the fault is no longer armed, so it completes there
        aborts, left = hb.persistent_probation()
        assert aborts == 1 and left is not None and 0 < left <= hb.PERSIST_RETRY_STEPS
        hb._PROBATION["retry_at"] = hb._PROBATION["steps"] + 2
        before = hb.LAUNCHES.get("lstm_fwd_persist", 0)
        mean2 = solver.sup_train_one_epoch(1, 1.0)
        solver.flush()
        assert hb.USE_PERSIST and hb.persistent_probation() == (1, None)
        assert hb.LAUNCHES.get("lstm_fwd_persist", 0) >= before + steps - 1, "the encoder is back on the persistent kernel"
        assert np.isfinite(mean2) and not hb.persist_aborted(dev)
    finally:
        hb.USE_PERSIST, hb.USE_PERSIST_DEC, hb.USE_PERSIST_DEC_BWD = flags
        hb.persist_clear_abort(dev)
