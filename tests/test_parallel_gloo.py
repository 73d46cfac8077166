"""Data-parallel logic on CPU: world_size-2 gloo ranks must reproduce the single-process gradients exactly
(SURVEY 8e): strided shard, global padded extents, local loss = -sum/(B_global*olength), ONE all-reduce of
the flat gradient buffer.  The compute is the CPU oracle; the sharding / flat-buffer / collective code is the
product's parallel.py: the three step kinds of the reference (supervised solver.py:375-378, semi-supervised
generator step 460-483 with its globally normalised auxiliary loss, judge step 288-291)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _setup_paths():
    for p in (ROOT, os.path.join(ROOT, "semi-supervised-asr_amd"), os.path.join(ROOT, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)


CFG = dict(synth.TINY)
ILENS = [11, 10, 9, 6, 5, 3]
YLENS = [4, 2, 3, 2, 3, 2]


def _grads_for(rank, world, tf_rate=1.0):
    from oracle import asr_oracle as O
    import parallel
    cfg = dict(CFG, labeldist=synth.labeldist(CFG["output_dim"], 12))
    sd = O.make_leaf_state(synth.e2e_weights(CFG, 11))
    names = O.unique_param_names(sd)
    buf = parallel.FlatBuffers([sd[n] for n in names])
    xs, ilens, ys = synth.batch(CFG["input_dim"], CFG["output_dim"], ILENS, YLENS, 13)
    xs_r, il_r, ys_r, info = parallel.shard_batch(xs, ilens, ys, rank, world)
    tl = O.padded_lengths(info["t_max"], CFG["enc_n_layers"], CFG["subsample"])
    np.random.seed(4)                                    # every rank draws the same tf sequence (F7)
    _, lp, _, _ = O.e2e_forward(sd, cfg, torch.from_numpy(np.ascontiguousarray(xs_r)), il_r,
                                [torch.from_numpy(y) for y in ys_r], tf_rate=tf_rate, total_length=tl,
                                olength_override=info["olength"])
    loss = parallel.local_loss(lp, info)
    buf.zero_grad()
    loss.backward()
    buf.collect()                                        # gradients -> flat buffer (FlatAdam.step does this itself)
    return buf, names, float(loss.detach())


def _model_fwd(sd, cfg):
    """E2E.forward's signature (as parallel.py calls it) on the oracle."""
    from oracle import asr_oracle as O

    def fwd(xs, ilens, ys=None, olength=None, **kw):
        xs_t = xs if torch.is_tensor(xs) else torch.from_numpy(np.ascontiguousarray(xs))
        return O.e2e_forward(sd, cfg, xs_t, ilens, ys, olength_override=olength, **kw)
    return fwd


UILENS = [12, 10, 9, 7, 4]                               # 5 unlabeled utterances over 2 ranks: uneven shards


def _ssl_for(rank, world):
    """Generator step of the semi-supervised training through parallel.ssl_local_loss -> (flat grads, scalars)."""
    from oracle import asr_oracle as O
    import parallel
    cfg = dict(CFG, labeldist=synth.labeldist(CFG["output_dim"], 12))
    sd = O.make_leaf_state(synth.e2e_weights(CFG, 11))
    jsd = {k: torch.from_numpy(v) for k, v in synth.lm_weights(synth.TINY_LM, 31).items()}
    names = O.unique_param_names(sd)
    buf = parallel.FlatBuffers([sd[n] for n in names])
    xs, ilens, ys = synth.batch(CFG["input_dim"], CFG["output_dim"], ILENS, YLENS, 13)
    uxs, uilens, _ = synth.batch(CFG["input_dim"], CFG["output_dim"], UILENS, [2] * len(UILENS), 41)

    def judge_probs(hyp):
        with torch.no_grad():
            return O.lm_forward(jsd, hyp, discrete_input=False, n_layers=2)[1]

    np.random.seed(9)
    loss, (unsup, sup) = parallel.ssl_local_loss(
        _model_fwd(sd, cfg), judge_probs, (xs, ilens, [torch.from_numpy(y) for y in ys]), (torch.from_numpy(uxs), uilens),
        rank, world, eos=2, unsup_weight=0.5, proportion=0.5, smooth=True, scaling=3.0,
        n_layers=CFG["enc_n_layers"], subsample=CFG["subsample"])
    buf.zero_grad()
    loss.backward()
    buf.set_aux([unsup, sup, loss])
    buf.allreduce_grads()                                # the step's single gradient collective (+ aux scalars)
    return buf, buf.aux[:3].tolist()


def _judge_for(rank, world):
    from oracle import asr_oracle as O
    import parallel
    jsd = {k: torch.tensor(v, requires_grad=True) for k, v in synth.lm_weights(synth.TINY_LM, 31).items()}
    buf = parallel.FlatBuffers(list(jsd.values()))
    rs = np.random.RandomState(33)
    ys = [torch.from_numpy(rs.randint(3, 9, size=(n,)).astype(np.int64)) for n in (6, 5, 4, 3, 3)]
    ld = synth.labeldist(9, 32)
    loss, avg = parallel.judge_local_loss(
        lambda y: O.lm_forward(jsd, y, discrete_input=True, n_layers=2, ls_weight=0.05, labeldist=ld),
        O.lm_masked_sum, ys, rank, world)
    buf.zero_grad()
    loss.backward()
    buf.set_aux([loss, avg])
    buf.allreduce_grads()
    return buf, buf.aux[:2].tolist()


def _worker_kind(rank, world, port, kind, out_dir):
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    buf, scalars = (_ssl_for if kind == "ssl" else _judge_for)(rank, world)
    if rank == 0:
        torch.save(dict(flat=buf.flat_g[:buf.total].clone(), scalars=scalars), os.path.join(out_dir, kind + ".pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["ssl", "judge"])
def test_ssl_and_judge_steps_two_ranks_equal_one_rank(tmp_path, kind):
    """The semi-supervised generator step (auxiliary loss normalised by the GLOBAL hypothesis-token count, uneven
    shards) and the judge step (normalised by the global sum of len+5): 2 ranks == 1 process, losses and gradients."""
    _setup_paths()
    port = _free_port()
    mp.spawn(_worker_kind, args=(2, port, kind, str(tmp_path)), nprocs=2, join=True)
    got = torch.load(os.path.join(str(tmp_path), kind + ".pt"))
    ref_buf, ref_scalars = (_ssl_for if kind == "ssl" else _judge_for)(0, 1)
    for a, b in zip(got["scalars"], ref_scalars):
        assert abs(a - b) <= 1e-6 * max(1.0, abs(b)), (got["scalars"], ref_scalars)
    ref = ref_buf.flat_g[:ref_buf.total]
    err = (got["flat"] - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert scale > 0 and err <= 1e-6 * max(scale, 1.0) + 1e-7, (err, scale)
    if kind == "ssl":
        # and the one-process value of the helper is the reference's own loss assembly (oracle.ssl_losses)
        from oracle import asr_oracle as O
        cfg = dict(CFG, labeldist=synth.labeldist(CFG["output_dim"], 12))
        sd = O.make_leaf_state(synth.e2e_weights(CFG, 11))
        jsd = {k: torch.from_numpy(v) for k, v in synth.lm_weights(synth.TINY_LM, 31).items()}
        xs, ilens, ys = synth.batch(CFG["input_dim"], CFG["output_dim"], ILENS, YLENS, 13)
        uxs, uilens, _ = synth.batch(CFG["input_dim"], CFG["output_dim"], UILENS, [2] * len(UILENS), 41)
        np.random.seed(9)
        sup, unsup = O.ssl_losses(sd, jsd, cfg, dict(n_layers=2), torch.from_numpy(xs), ilens,
                                  [torch.from_numpy(y) for y in ys], torch.from_numpy(uxs), uilens, 0.5)
        assert abs(float(unsup) - ref_scalars[0]) < 1e-6 and abs(float(sup) - ref_scalars[1]) < 1e-6


def test_empty_shard_keeps_rng_aligned():
    """More ranks than utterances: the idle rank returns no loss but consumes the same teacher-forcing draws."""
    _setup_paths()
    import parallel
    xs, ilens, ys = synth.batch(CFG["input_dim"], CFG["output_dim"], [7, 5], [3, 2], 13)
    np.random.seed(3)
    assert parallel.sup_local_loss(None, xs, ilens, [torch.from_numpy(y) for y in ys], 0.5, 2, 3, 2, [2, 2]) is None
    after_idle = np.random.random_sample()
    np.random.seed(3)
    for _ in range(4):                                   # olength = max label length + 1
        np.random.random_sample()
    assert after_idle == np.random.random_sample()


def _worker(rank, world, port, tf_rate, out_dir):
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    buf, names, loss = _grads_for(rank, world, tf_rate)
    buf.allreduce_grads()                                # the single collective of the step
    t = torch.tensor([loss], dtype=torch.float64)
    dist.all_reduce(t)
    if rank == 0:
        torch.save(dict(flat=buf.flat_g.clone(), loss=float(t.item())), os.path.join(out_dir, "dp.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("tf_rate", [1.0, 0.5])
def test_two_ranks_equal_one_rank(tmp_path, tf_rate):
    _setup_paths()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, tf_rate, str(tmp_path)), nprocs=2, join=True)
    got = torch.load(os.path.join(str(tmp_path), "dp.pt"))
    ref_buf, names, ref_loss = _grads_for(0, 1, tf_rate)
    assert abs(got["loss"] - ref_loss) < 1e-6 * max(1.0, abs(ref_loss))
    err = (got["flat"] - ref_buf.flat_g).abs().max().item()
    scale = ref_buf.flat_g.abs().max().item()
    assert err <= 1e-6 * max(scale, 1.0) + 1e-7, (err, scale)


def _overlap_grads(rank, world, n_utts, overlap, then_disable=False):
    """Supervised step on the first n_utts utterances of the test batch; with `overlap` the flat buffer is cut into small
    buckets whose all-reduces are issued from inside the backward pass (FlatBuffers.enable_overlap)."""
    from oracle import asr_oracle as O
    import parallel
    cfg = dict(CFG, labeldist=synth.labeldist(CFG["output_dim"], 12))
    sd = O.make_leaf_state(synth.e2e_weights(CFG, 11))
    names = O.unique_param_names(sd)
    parallel.FlatBuffers.BUCKET_FLOATS = 700              # tiny model: several buckets
    buf = parallel.FlatBuffers([sd[n] for n in names])
    assert len(buf.buckets) >= 3 and buf.buckets[0][3] == buf.total and buf.buckets[-1][2] == 0
    if overlap:
        buf.enable_overlap()
        assert buf.overlap == (world > 1)
    xs, ilens, ys = synth.batch(CFG["input_dim"], CFG["output_dim"], ILENS[:n_utts], YLENS[:n_utts], 13)
    ys_t = [torch.from_numpy(y) for y in ys]
    np.random.seed(4)
    loss = parallel.sup_local_loss(_model_fwd(sd, cfg), xs, ilens, ys_t, 1.0, rank, world, CFG["enc_n_layers"], CFG["subsample"])
    buf.zero_grad()
    if loss is not None:
        loss.backward()
        if overlap and world > 1:
            assert buf._issued >= 1, "no bucket was issued from inside the backward pass"
    buf.set_aux([loss if loss is not None else 0.0])
    buf.allreduce_grads()
    assert all(p.grad is not None and p.grad.data_ptr() == buf._view(i).data_ptr() for i, p in enumerate(buf.params))
    if then_disable:
        # bench.py's first fallback: back to ONE collective on the same buffers (hooks removed), same gradients
        first = buf.flat_g.clone()
        buf.disable_overlap()
        assert not buf.overlap and not buf._hooks
        np.random.seed(4)
        loss = parallel.sup_local_loss(_model_fwd(sd, cfg), xs, ilens, ys_t, 1.0, rank, world, CFG["enc_n_layers"], CFG["subsample"])
        buf.zero_grad()
        if loss is not None:
            loss.backward()
            assert buf._issued == 0
        buf.set_aux([loss if loss is not None else 0.0])
        buf.allreduce_grads()
        assert torch.allclose(buf.flat_g, first, rtol=0, atol=1e-7 * max(1.0, float(first.abs().max())))
    return buf


def _worker_overlap(rank, world, port, n_utts, out_dir):
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    buf = _overlap_grads(rank, world, n_utts, True)
    # a second step on the same buffers: the per-step bucket state is reset by zero_grad()
    buf2 = _overlap_grads(rank, world, n_utts, True, then_disable=True)
    if rank == 0:
        torch.save(dict(flat=buf.flat_g.clone(), flat2=buf2.flat_g.clone()), os.path.join(out_dir, "ov.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_utts", [(2, 6), (3, 2)])
def test_overlapped_bucket_allreduce_equals_one_process(tmp_path, world, n_utts):
    """The bucketed gradient exchange issued from inside the backward pass (FlatBuffers.enable_overlap: buckets counted from
    the end of the flat buffer, issued in a fixed order) against the one-process gradients: 2 ranks, and 3 ranks of which
    one has an empty shard - it runs no backward and issues the same sequence of collectives in allreduce_grads()."""
    _setup_paths()
    port = _free_port()
    mp.spawn(_worker_overlap, args=(world, port, n_utts, str(tmp_path)), nprocs=world, join=True)
    got = torch.load(os.path.join(str(tmp_path), "ov.pt"))
    import parallel
    keep = parallel.FlatBuffers.BUCKET_FLOATS
    try:
        ref = _overlap_grads(0, 1, n_utts, False).flat_g
    finally:
        parallel.FlatBuffers.BUCKET_FLOATS = keep
    scale = ref.abs().max().item()
    for k in ("flat", "flat2"):
        err = (got[k] - ref).abs().max().item()
        assert scale > 0 and err <= 1e-6 * max(scale, 1.0) + 1e-7, (k, err, scale)


def test_shard_is_strided_and_sorted():
    _setup_paths()
    import parallel
    xs, ilens, ys = synth.batch(CFG["input_dim"], CFG["output_dim"], ILENS, YLENS, 13)
    seen = []
    for r in range(3):
        xs_r, il_r, ys_r, info = parallel.shard_batch(xs, ilens, ys, r, 3)
        assert il_r == sorted(il_r, reverse=True)
        assert xs_r.shape[1] == max(ILENS) and info["b_global"] == len(ILENS) and info["olength"] == max(YLENS) + 1
        seen += parallel.shard_indices(len(ilens), r, 3)
    assert sorted(seen) == list(range(len(ILENS)))


def test_flat_buffers_alias_params_and_grads():
    _setup_paths()
    import parallel
    lin = torch.nn.Linear(5, 3)
    shared = torch.nn.Parameter(torch.ones(7))
    buf = parallel.FlatBuffers([lin.weight, lin.bias, shared, shared])
    assert len(buf.params) == 3 and buf.total % 4 == 0
    buf.zero_grad()
    (lin(torch.ones(2, 5)).sum() + (shared * 2).sum()).backward()
    assert torch.equal(lin.bias.grad, torch.full((3,), 2.0))
    buf.collect()
    assert lin.bias.grad.data_ptr() == buf.flat_g.data_ptr() + 4 * buf.offsets[1]
    o = buf.offsets[2]
    assert torch.equal(buf.flat_g[o:o + 7], torch.full((7,), 2.0))
    buf.flat_p[buf.offsets[1]:buf.offsets[1] + 3] = 5.0
    assert torch.equal(lin.bias.data, torch.full((3,), 5.0))
    # aux slots ride behind the gradients and survive zero_grad / collect
    buf.set_aux([torch.tensor(1.5), 2.0])
    buf.zero_grad()
    buf.collect()
    assert buf.flat_g.numel() == buf.total + buf.NAUX and buf.aux.tolist() == [1.5, 2.0, 0.0, 0.0]


class _SgdOpt(object):
    """Stand-in for FlatAdam on CPU (its apply() is a HIP kernel): the same zero_grad / buf / reduce / apply surface around
    the product's FlatBuffers, with a plain SGD update."""

    def __init__(self, params, lr):
        import parallel
        self.buf, self.lr, self.applied, self.t = parallel.FlatBuffers(params), lr, 0, 0

    def zero_grad(self):
        self.buf.zero_grad()

    def reduce(self):
        self.buf.allreduce_grads()

    def apply(self, skip_if=None):
        """skip_if: a 1-element tensor; not zero = the update is a no-op (FlatAdam: checked by the kernel on the device)
        that still counts as a step until unapply() takes it back."""
        self.t += 1
        if skip_if is not None and float(skip_if) != 0.0:
            return
        self.applied += 1
        self.buf.flat_p -= self.lr * self.buf.flat_g[:self.buf.total]

    def unapply(self, n=1):
        self.t -= n


def _dp_steps(rank, world, abort_rank, pipelined=0):
    """Two optimiser steps through parallel.dp_step (what Solver._dp_step runs) with tf_rate 0.5 - the teacher-forcing
    draws come from the numpy stream - where rank `abort_rank` reports an aborted persistent kernel (latch set, loss
    poisoned) in the FIRST attempt of the FIRST step.  pipelined = depth > 0: three steps through parallel.DpPipeline
    (the Solver's default under data parallelism) instead."""
    from oracle import asr_oracle as O
    import parallel
    cfg = dict(CFG, labeldist=synth.labeldist(CFG["output_dim"], 12))
    sd = O.make_leaf_state(synth.e2e_weights(CFG, 11))
    opt = _SgdOpt([sd[n] for n in O.unique_param_names(sd)], 0.05)
    xs, ilens, ys = synth.batch(CFG["input_dim"], CFG["output_dim"], ILENS, YLENS, 13)
    ys_t = [torch.from_numpy(y) for y in ys]
    state = dict(latch=0.0, fast=True, calls=0, left=[])

    def make_loss():
        state["calls"] += 1
        loss = parallel.sup_local_loss(_model_fwd(sd, cfg), xs, ilens, ys_t, 0.5, rank, world, CFG["enc_n_layers"], CFG["subsample"])
        if state["fast"] and rank == abort_rank and state["calls"] == 1:
            state["latch"] = 1.0                          # what an aborting persistent kernel leaves behind:
            loss = loss * float("nan")                    # the sticky latch and NaN-poisoned outputs
        return loss, [loss]

    def leave(n_ranks):
        state["left"].append(n_ranks)
        state["fast"], state["latch"] = False, 0.0        # hb.disable_persistent: off the fast path, latch cleared

    np.random.seed(21)
    if pipelined:
        pipe = parallel.DpPipeline(pipelined, lambda: state["latch"], leave)
        recs = [pipe.step(make_loss, opt, 1) for _ in range(3)]
        state["outstanding"] = [r["values"] is None for r in recs]
        pipe.flush()
        assert not pipe.pending and opt.t == 3
        return opt, [r["values"][0] for r in recs], state
    losses = [parallel.dp_step(make_loss, opt, 1, lambda: state["latch"], leave)[0] for _ in range(2)]
    return opt, losses, state


def _worker_dp_abort(rank, world, port, out_dir):
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    opt, losses, state = _dp_steps(rank, world, abort_rank=1)
    # every rank took the same decision: one repeat of step 1 (3 forward passes for 2 steps), 2 updates applied
    assert state["calls"] == 3 and opt.applied == 2 and state["left"] == [1] and not state["fast"], (rank, state, opt.applied)
    torch.save(dict(p=opt.buf.flat_p.clone(), losses=losses), os.path.join(out_dir, "dpabort%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_step_coordinated_fallback_after_an_abort_on_one_rank(tmp_path):
    """An aborted persistent kernel on rank 1 only (latch set, NaN loss): the latch rides in the step's all-reduce, BOTH
    ranks discard the attempt, restore the numpy stream, leave the fast path and repeat - the weights after two steps
    equal the one-process run that never aborted, on both ranks."""
    _setup_paths()
    port = _free_port()
    mp.spawn(_worker_dp_abort, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    ref_opt, ref_losses, ref_state = _dp_steps(0, 1, abort_rank=-1)
    assert ref_state["calls"] == 2 and ref_state["left"] == []
    ref = ref_opt.buf.flat_p
    for r in range(2):
        got = torch.load(os.path.join(str(tmp_path), "dpabort%d.pt" % r))
        assert all(np.isfinite(v) for v in got["losses"])
        for a, b in zip(got["losses"], ref_losses):
            assert abs(a - b) <= 1e-6 * max(1.0, abs(b)), (got["losses"], ref_losses)
        err = (got["p"] - ref).abs().max().item()
        assert err <= 1e-6 * max(1.0, ref.abs().max().item()), (r, err)


def _worker_dp_pipeline(rank, world, port, out_dir):
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    opt, losses, state = _dp_steps(rank, world, abort_rank=1, pipelined=1)
    # step 1 aborted on rank 1, step 2 was enqueued behind it with the latch still set (both skipped on BOTH ranks), the
    # record of step 1 was read after step 2: both repeated off the fast path, then step 3: 5 forward passes, 3 updates
    assert state["calls"] == 5 and opt.applied == 3 and state["left"] == [1] and not state["fast"], (rank, state, opt.applied)
    assert state["outstanding"] == [False, False, True]      # the last step's record is read by flush()
    torch.save(dict(p=opt.buf.flat_p.clone(), losses=losses), os.path.join(out_dir, "dppipe%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_pipeline_skips_on_every_rank_and_repeats_one_step_late(tmp_path):
    """The data-parallel step without a host wait: the update is predicated on the all-reduced latch, the host reads a
    step's record while the next one runs.  Rank 1 aborts in step 1: neither step 1 nor step 2 (enqueued behind it) is
    applied on either rank; both ranks repeat them in order from the numpy stream of step 1 - weights and losses after
    three steps equal the one-process run that never aborted."""
    _setup_paths()
    port = _free_port()
    mp.spawn(_worker_dp_pipeline, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    ref_opt, ref_losses, ref_state = _dp_steps(0, 1, abort_rank=-1, pipelined=1)
    assert ref_state["calls"] == 3 and ref_state["left"] == [] and ref_opt.applied == 3
    ref = ref_opt.buf.flat_p
    for r in range(2):
        got = torch.load(os.path.join(str(tmp_path), "dppipe%d.pt" % r))
        assert all(np.isfinite(v) for v in got["losses"])
        for a, b in zip(got["losses"], ref_losses):
            assert abs(a - b) <= 1e-6 * max(1.0, abs(b)), (got["losses"], ref_losses)
        err = (got["p"] - ref).abs().max().item()
        assert err <= 1e-6 * max(1.0, ref.abs().max().item()), (r, err)


def test_dp_pipeline_depth_two_repeats_everything_enqueued_behind_the_abort():
    """One process, two records outstanding: the abort of step 1 is found after steps 2 and 3 have been enqueued (all three
    skipped: the latch is sticky), the three are repeated in order from the numpy stream of step 1 - same weights and
    losses as the run that never aborted."""
    _setup_paths()
    ref_opt, ref_losses, ref_state = _dp_steps(0, 1, abort_rank=-1, pipelined=2)
    opt, losses, state = _dp_steps(0, 1, abort_rank=0, pipelined=2)
    assert ref_state["calls"] == 3 and state["calls"] == 6 and opt.applied == 3 and state["left"] == [1], (state, opt.applied)
    assert ref_state["outstanding"] == [False, True, True]       # step 3's enqueue resolved step 1; two records stay in flight
    assert state["outstanding"] == [False, False, False]         # ... and with the abort it resolved all three (the replay is synchronous)
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 1e-6 * max(1.0, abs(b)), (losses, ref_losses)
    assert (opt.buf.flat_p - ref_opt.buf.flat_p).abs().max().item() <= 1e-6 * max(1.0, ref_opt.buf.flat_p.abs().max().item())


def test_dp_step_raises_when_the_repeat_fails_too():
    _setup_paths()
    import parallel
    lin = torch.nn.Linear(3, 2)
    opt = _SgdOpt(list(lin.parameters()), 0.1)
    before = opt.buf.flat_p.clone()
    with pytest.raises(RuntimeError, match="nothing was applied"):
        parallel.dp_step(lambda: ((lin(torch.ones(1, 3)).sum()),) * 1 + ([],), opt, 0, lambda: 1.0, lambda n: None)
    assert opt.applied == 0 and torch.equal(opt.buf.flat_p, before)
