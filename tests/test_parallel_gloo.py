"""Data-parallel logic on CPU: world_size-2 gloo ranks must reproduce the single-process gradients exactly
(SURVEY 8e): strided shard, global padded extents, local loss = -sum/(B_global*olength), ONE all-reduce of
the flat gradient buffer.  The compute is the CPU oracle; the sharding / flat-buffer / collective code is the
product's parallel.py."""
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
