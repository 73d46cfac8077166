"""Utterance-level data parallelism + the flat-buffer optimiser.

The reference has no distributed code; this is the new component the north star
asks for: one process per GPU, the global minibatch sharded by utterance, ONE all-reduce (RCCL over
xGMI; backend "nccl" on ROCm) of a single flat fp32 gradient buffer per step, issued after backward
and before the clip, then an identical fused clip+Adam step on every rank.

Exact-parity rules under sharding are host-side only:
  * strided shard (rank r gets global rows r, r+W, ...) keeps every shard length-sorted;
  * every shard is padded to the GLOBAL T_max and decodes the GLOBAL olength (the unmasked softmax
    and unmasked mean make results depend on both);
  * local loss = -sum_local(log_probs) / (B_global * olength_global), so the all-reduced SUM of
    gradients equals the single-process gradient.
"""
import math

import torch
import torch.distributed as dist


# ------------------------------------------------------------------------------ sharding (host)
def shard_indices(n, rank, world):
    """Strided assignment: rows rank, rank+world, ... of a length-sorted global batch."""
    return list(range(rank, n, world))


def shard_batch(xs, ilens, ys, rank, world):
    """(xs [B,T,D] zero-padded to the global T_max, ilens desc, ys list) -> this rank's rows.
    xs keeps the global padded length; returns (xs_r, ilens_r, ys_r, info) where info carries the
    global constants every rank needs (B_global, T_max, olength)."""
    idx = shard_indices(len(ilens), rank, world)
    info = dict(b_global=len(ilens), t_max=int(max(ilens)),
                olength=(max(int(y.shape[0]) for y in ys) + 1) if ys is not None else None)
    xs_r = xs[idx]
    return xs_r, [ilens[i] for i in idx], ([ys[i] for i in idx] if ys is not None else None), info


def local_loss(log_probs, info):
    """-sum over this shard / (B_global * olength_global)  (solver.py:377 is the W=1 case)."""
    return -log_probs.sum() / float(info["b_global"] * log_probs.shape[1])


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


# ------------------------------------------------------------------------------ flat buffers
class FlatBuffers(object):
    """Re-home every parameter (and its .grad) of a module into two flat fp32 buffers so that the
    gradient exchange is one collective and the optimiser one kernel.  Shared parameters (the
    attention module appears twice in E2E) are stored once.  Offsets are padded to 4
    floats so each view stays 16-byte aligned for the kernels."""

    def __init__(self, params):
        self.params = []
        seen = set()
        for p in params:
            if id(p) not in seen and p.requires_grad:
                seen.add(id(p))
                self.params.append(p)
        self.offsets = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4
        self.total = off
        dev = self.params[0].device
        self.flat_p = torch.zeros(off, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(off, device=dev, dtype=torch.float32)
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            self.flat_p[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat_p[o:o + n].view_as(p.data)
            p.grad = self.flat_g[o:o + n].view_as(p.data)

    def zero_grad(self):
        """Detach every .grad: autograd then stores each gradient by reference instead of launching one add kernel per
        parameter into the flat buffer; collect() gathers them with a single multi-tensor copy before the step."""
        for p in self.params:
            p.grad = None

    def _view(self, i):
        p, o = self.params[i], self.offsets[i]
        return self.flat_g[o:o + p.numel()].view_as(p.data)

    def collect(self):
        """Move the gradients autograd produced into the flat buffer and re-attach .grad to its views."""
        dst, src, missing = [], [], False
        for i, p in enumerate(self.params):
            v = self._view(i)
            if p.grad is None:
                missing = True
            elif p.grad.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(p.grad)
        if missing:
            self.flat_g.zero_()
        if dst:
            torch._foreach_copy_(dst, src)
        for i, p in enumerate(self.params):
            p.grad = self._view(i)

    def allreduce_grads(self, group=None):
        """THE collective of the step: one SUM all-reduce over the flat gradient buffer."""
        self.collect()
        if world() > 1:
            dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM, group=group)


# ------------------------------------------------------------------------------ optimiser
class FlatAdam(object):
    """clip_grad_norm_ + Adam(amsgrad, L2 weight decay) (solver.py:152-153,384-385) fused into two
    HIP kernels over the flat buffers (asr_sumsq_f32, asr_adam_clip_f32); no host sync in step().
    state_dict()/load_state_dict() use torch.optim.Adam's schema so `.opt` checkpoints interchange."""

    def __init__(self, module_or_params, lr, weight_decay=0.0, amsgrad=False, betas=(0.9, 0.999), eps=1e-8,
                 max_grad_norm=None):
        params = module_or_params.parameters() if hasattr(module_or_params, "parameters") else module_or_params
        self.buf = FlatBuffers(list(params))
        self.param_groups = [dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad,
                                  params=list(range(len(self.buf.params))))]
        self.max_grad_norm = max_grad_norm
        dev = self.buf.flat_p.device
        self.m = torch.zeros_like(self.buf.flat_p)
        self.v = torch.zeros_like(self.buf.flat_p)
        self.vmax = torch.zeros_like(self.buf.flat_p) if amsgrad else None
        self.gnorm_sq = torch.zeros(1, device=dev, dtype=torch.float32)
        self.t = 0

    def zero_grad(self):
        self.buf.zero_grad()

    def step(self, max_grad_norm=None, group=None):
        """all-reduce (if distributed) -> global grad norm -> clip + Adam.  Returns the device scalar
        holding ||g||^2 (read it with .item() only if you need the number)."""
        import hip_backend as hb
        clip = self.max_grad_norm if max_grad_norm is None else max_grad_norm
        self.buf.allreduce_grads(group)                # gathers the gradients into the flat buffer first
        g = self.param_groups[0]
        self.t += 1
        b1, b2 = g["betas"]
        lib = hb.load()
        n = self.buf.total
        gptr = None
        if clip is not None:
            self.gnorm_sq.zero_()
            hb.check(lib.asr_sumsq_f32(n, hb.ptr(self.buf.flat_g), hb.ptr(self.gnorm_sq), hb.stream()),
                     "asr_sumsq_f32")
            gptr = hb.ptr(self.gnorm_sq)
        hb.check(lib.asr_adam_clip_f32(n, hb.ptr(self.buf.flat_p), hb.ptr(self.buf.flat_g), hb.ptr(self.m),
                                       hb.ptr(self.v), hb.ptr(self.vmax), gptr,
                                       float(clip if clip is not None else 0.0), float(g["lr"]), float(b1),
                                       float(b2), float(g["eps"]), float(g["weight_decay"]),
                                       1.0 - b1 ** self.t, 1.0 - b2 ** self.t, hb.stream()), "asr_adam_clip_f32")
        return self.gnorm_sq

    # ---- torch.optim.Adam-compatible (de)serialisation
    def state_dict(self):
        state = {}
        if self.t > 0:
            for i, (p, o) in enumerate(zip(self.buf.params, self.buf.offsets)):
                n = p.numel()
                ent = dict(step=torch.tensor(float(self.t)), exp_avg=self.m[o:o + n].view_as(p).clone(),
                           exp_avg_sq=self.v[o:o + n].view_as(p).clone())
                if self.vmax is not None:
                    ent["max_exp_avg_sq"] = self.vmax[o:o + n].view_as(p).clone()
                state[i] = ent
        return dict(state=state, param_groups=[dict(self.param_groups[0])])

    def load_state_dict(self, sd):
        grp = sd["param_groups"][0]
        for k in ("lr", "betas", "eps", "weight_decay"):
            if k in grp:
                self.param_groups[0][k] = grp[k]
        for i, ent in sd.get("state", {}).items():
            i = int(i)
            p, o = self.buf.params[i], self.buf.offsets[i]
            n = p.numel()
            self.m[o:o + n].copy_(ent["exp_avg"].reshape(-1))
            self.v[o:o + n].copy_(ent["exp_avg_sq"].reshape(-1))
            if self.vmax is not None and "max_exp_avg_sq" in ent:
                self.vmax[o:o + n].copy_(ent["max_exp_avg_sq"].reshape(-1))
            self.t = int(float(ent["step"]))


def init_distributed():
    """One process per GPU (torch.distributed.run env).  Returns (rank, world, local_rank)."""
    import os
    w = int(os.environ.get("WORLD_SIZE", "1"))
    if w <= 1:
        return 0, 1, int(os.environ.get("LOCAL_RANK", "0"))
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if torch.cuda.is_available():
        ndev = max(1, torch.cuda.device_count())
        local = local % ndev                                    # rehearsal: several ranks may share one card
        torch.cuda.set_device(local)
        if w > ndev:
            # The persistent kernels need all 256 CUs of a device co-resident (one workgroup per CU).  Two processes
            # sharing a card could each get part of it and starve each other until the bounded spins abort, so a
            # shared-card rehearsal takes the per-step kernels.
            import hip_backend as hb
            hb.USE_PERSIST = hb.USE_PERSIST_DEC = hb.USE_PERSIST_DEC_BWD = False
        backend = os.environ.get("ASR_DIST_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
    else:
        backend = "gloo"
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=w)
    return rank, w, local
