"""torch.autograd.Function wrappers over the C-ABI kernels (hip_backend).

One Function per *sequence-level* operator so that the sequential chains (encoder
recurrence, decoder loop) never bounce through Python/autograd per time step:

    linear            y = relu?(x W^T + b)                       -> asr_gemm_f32
    lstm_layer        whole (bi)LSTM layer, time-major            -> asr_gemm_f32 + asr_lstm_seq_*
    pyramid_concat    pair-concat (+dropout mask)                 -> asr_pyramid_concat_*
    decoder_sequence  all decoder steps incl. attention           -> asr_dec_* (+ GEMMs)

Workspaces.  The per-time-step launch chains are replayed as HIP graphs whose kernel arguments are baked
(see csrc/graphs.h), so the chain buffers must keep their addresses from step to step.  Each chain op
therefore leases a preallocated, shape-keyed workspace from a pool for the span forward -> end of backward
(`_Lease`); a second concurrent user of the same shape (e.g. the two model passes of the SSL step) simply gets
another instance.  With autograd disabled (validation / greedy decoding) plain fresh tensors are used.
Outputs that alias a workspace (`y`) are only valid until that op's backward has run — the model consumes
them immediately; double backward / retain_graph through these ops is not supported.

GPU only; see hip_backend for the no-fallback rule.
"""
import ctypes
import os

import numpy as np
import torch

import hip_backend as hb


def gate_perm(H, device):
    """Row permutation torch (gate-major i,f,g,o) -> gate-interleaved (unit*4+gate)."""
    return torch.arange(4 * H, device=device).view(4, H).t().reshape(-1)


def gate_unperm(H, device):
    return torch.arange(4 * H, device=device).view(H, 4).t().reshape(-1)


# -------------------------------------------------------------------------------------- workspace pool
class _Lease(object):
    """Exclusive use of one workspace (a dict of tensors) until release() / garbage collection."""

    def __init__(self, free_list, ws):
        self._free, self.ws = free_list, ws

    def release(self):
        if self.ws is not None:
            self._free.append(self.ws)
            self.ws = None

    __del__ = release


class _Pool(object):
    def __init__(self):
        self.free = {}

    def acquire(self, key, factory):
        lst = self.free.setdefault(key, [])
        return _Lease(lst, lst.pop() if lst else factory())


_POOL = _Pool()


# -------------------------------------------------------------------------------------- accumulator arena
class _Arena(object):
    """Every buffer of a train step that has to START FROM ZERO - split-K GEMM outputs (their partial products meet in
    atomics; the outputs of ops.linear among them), bias-gradient sums, the accumulators of the sequence operators' backward
    passes, the loss scalar - cut from ONE buffer that
    ONE fill zeroes at the start of the step (Solver._step opens the scope: `with ops.step_arena(device)`), instead of a
    zero pass or a memset in front of each (39 fills per cfg-2 step, 0.2 ms).  A slice is valid until the next scope opens;
    outside a scope, and for what does not fit yet (the buffer grows to the step's need at the next scope), take() returns
    None and the caller falls back to its own zero fill.

    LIFETIME.  A slice is raw storage of the arena (set_ on the untyped storage: autograd cannot see that slices of
    different steps overlap), so everything a step produced inside its scope - the outputs of ops.linear (the logits and
    log-probs of E2E.forward among them), the weight gradients in .grad - is valid UNTIL THE NEXT SCOPE OPENS and is zeroed
    then.  Solver._step keeps to that: its closures hand back scalars only, which are copied to the step's host record
    inside the scope.  Code that wants to keep a tensor of a step (logging logits, comparing gradients across steps)
    clones it before the next step, or runs the model outside a scope (tests, validation: take() returns None there).

    INVARIANT.  Everything at and behind the cursor is zero: begin() zeroes what the previous scope used, nothing else was
    ever written.  A scope opened inside another one (Solver._recover, from the end of the step that found the abort) relies
    on it; ASR_ARENA_DEBUG=1 checks it at every begin() (a device sync: debugging only)."""

    def __init__(self):
        self.buf, self.cursor, self.want, self.active = None, 0, 0, False

    def begin(self, device):
        device = torch.device(device)
        if self.buf is None or self.buf.device != device or self.buf.numel() < self.want:
            # (first step, or the last step asked for more than there is: a new, larger buffer - zeros already)
            self.buf = torch.zeros(int(self.want * 1.25) + 4096, device=device, dtype=torch.float32) if self.want > 0 else None
        elif self.cursor > 0:
            self.buf[:self.cursor].zero_()               # what the previous step used: ONE fill
        if _ARENA_DEBUG and self.buf is not None:
            assert float(self.buf.abs().max()) == 0.0, "step arena: a value survived behind the cursor"
        self.cursor, self.want, self.active = 0, 0, True

    def end(self):
        self.active = False

    def take(self, shape, device):
        if not self.active:
            return None
        n = 1
        for v in shape:
            n *= int(v)
        n4 = (n + 63) // 64 * 64                         # slices stay 256-byte aligned (GEMM operands live here)
        self.want += n4
        if self.buf is None or self.buf.device != torch.device(device) or self.cursor + n4 > self.buf.numel():
            return None
        # a tensor of its own over the buffer's storage, not a view of the buffer: views share ONE version counter, and an
        # in-place torch op on any slice (index_add_ into an accumulator) would then invalidate every slice that an
        # autograd node saved for its backward (the outputs of ops.linear live here)
        shape = tuple(int(v) for v in shape)
        strides, acc = [], 1
        for v in reversed(shape):
            strides.append(acc)
            acc *= v
        out = torch.empty(0, device=self.buf.device, dtype=torch.float32).set_(
            self.buf.untyped_storage(), self.buf.storage_offset() + self.cursor, shape, tuple(reversed(strides)))
        self.cursor += n4
        return out


_ARENA_DEBUG = os.environ.get("ASR_ARENA_DEBUG", "0") == "1"
_ARENA = _Arena()
_LINEAR_ARENA = os.environ.get("ASR_LINEAR_ARENA", "1") != "0"      # measurement: the outputs of ops.linear from the arena


class step_arena(object):
    """Scope of one train step (forward + backward + optimiser): see _Arena."""

    def __init__(self, device):
        self.device = device

    def __enter__(self):
        if torch.device(self.device).type == "cuda":
            _ARENA.begin(self.device)
        return _ARENA

    def __exit__(self, *exc):
        _ARENA.end()
        return False


def zeros_acc(shape, device):
    """A zero-initialised accumulator: a slice of the step's arena when there is one, else a fresh zero tensor."""
    t = _ARENA.take(shape, device)
    return t if t is not None else torch.zeros(*shape, device=device, dtype=torch.float32)


# -------------------------------------------------------------------------------------- weight gradients beside the chains
class _SideStream(object):
    """A batch of <= 8 utterances keeps every persistent kernel of the step (LSTM, decoder, judge) on four of the eight XCDs
    (hb.idle_xcd_mask).  The weight-gradient products of the backward pass are off its critical path - dG -> dX -> the
    recurrence of the layer below is - so they go to a SIDE stream, onto the idle XCDs, by workgroups that fit beside a
    persistent one (hb.gemm_side / asr_gemm_side_f32), and run under the recurrence of the layer below.
      defer(dev, launch, keep)   inside a backward function: queue products for the side stream (see there).  The first one
                         of a backward pass registers the join with the autograd engine: when the pass ends, the stream the
                         backward ran on waits for the side stream - code that reads .grad afterwards (the optimiser, a
                         test) finds every gradient complete without knowing about any of this.
      flush()            in front of a recurrence: start what is queued, beside it.
      join_now()         flush + the wait at once (a node that consumes side results inside the pass: _LstmPack.backward).
    Outputs are zeroed on the main stream BEFORE defer() (the side products add into them with atomics).  Off: ASR_SIDE_GEMM=0, the
    fp32-input MFMA arithmetic (no such kernel), gradients exchanged from inside the backward pass (dp_overlap: its hooks
    read a gradient as soon as autograd has it)."""

    def __init__(self):
        self.enabled = os.environ.get("ASR_SIDE_GEMM", "1") != "0"
        self.streams, self.active, self.mask_hint = {}, None, 0
        self.uses = {}                     # weight data_ptr -> forward passes since the last join (see _Linear)
        self.deferred = []                 # (launch closure, temporaries) waiting for the next recurrence
        self.task = -1                     # the autograd graph task (backward pass) `active` belongs to
        self.launches = 0

    def mask_for(self, nbatch):
        return hb.idle_xcd_mask(nbatch) if self.enabled else 0

    def usable(self, mask):
        return bool(mask) and self.enabled and (hb.current_arith() & 0xff) != hb.ARITH_F32

    def defer(self, dev, launch, keep):
        """Inside a backward function, once the operands are final on the current stream: queue `launch` (a closure that
        enqueues the products on whatever stream is current) for the side stream.  It is NOT started here - right behind a
        recurrence the main stream runs the critical dX products, and side workgroups on half the chip's CUs would starve
        them (measured: a [5 248, 2 048] x [2 048, 512] dX product 394 us instead of 106) - but when the NEXT recurrence is
        about to be enqueued (flush(), called by _LstmLayer.backward in front of its chain kernel), or when the pass ends.
        `keep`: the tensors the products read or write that are temporaries of the caller (kept alive until the launch, then
        handed to record_stream: the allocator must not give their memory to the main stream's next allocation - a fresh,
        ZEROED ticket counter, say - while the side stream still runs)."""
        main = torch.cuda.current_stream(dev)
        st = self.streams.get(dev.index)
        if st is None:
            st = self.streams[dev.index] = torch.cuda.Stream(device=dev)
        task = torch._C._current_graph_task_id()
        if self.active is not None and task != self.task:
            # a pass that never ended (an exception inside backward): its join never ran - what it left is dropped, the
            # side stream is drained, and this pass registers its own join
            self.active[0].wait_stream(self.active[1])
            del self.deferred[:]
            self.active = None
            self.uses.clear()
        if self.active is None:
            self.active, self.task = (main, st), task
            torch.autograd.Variable._execution_engine.queue_callback(self.join)
        self.deferred.append((launch, keep))
        self.launches += 1

    def flush(self):
        """Start what has been deferred: the side stream waits for everything enqueued on the main stream so far."""
        if self.active is None or not self.deferred:
            return
        if self.task != torch._C._current_graph_task_id() and torch._C._current_graph_task_id() != -1:
            return                         # (leftovers of a pass that never ended: defer() of THIS pass will drop them)
        main, st = self.active
        ev = torch.cuda.Event()
        ev.record(main)
        st.wait_event(ev)
        with torch.cuda.stream(st):
            for launch, keep in self.deferred:
                launch()
                for t in keep:
                    t.record_stream(st)
        del self.deferred[:]

    def join_now(self):
        if self.active is not None:
            self.flush()
            self.active[0].wait_stream(self.active[1])

    def join(self):
        self.join_now()
        self.active = None
        self.uses.clear()


_SIDE = _SideStream()


def _side_product(a, b, out, queue, mask):
    code = hb.current_arith()              # (the arithmetic of the node that queued it, not of whoever flushes)
    return lambda: hb.gemm_side(a, b, out, queue, mask, trans_a=True, arith=code)


def _gemm_acc(A, B, trans_a=False, trans_b=False, shape=None):
    """A product into a fresh output that no epilogue follows (the weight gradients): inside a step's arena the output is a
    pre-zeroed slice and the library accumulates into it - no zero pass in front of a split-K product."""
    out = _ARENA.take(shape, A.device) if shape is not None else None
    if out is None:
        return hb.gemm(A, B, trans_a=trans_a, trans_b=trans_b)
    return hb.gemm(A, B, trans_a=trans_a, trans_b=trans_b, out=out, accumulate=True)


def _colsum_acc(X):
    out = _ARENA.take((X.shape[1],), X.device)
    return hb.colsum(X) if out is None else hb.colsum(X, out=out, accumulate=True)


# --------------------------------------------------------------------------------------
def _with_saved_arith(bwd):
    """The backward of a sequence operator runs under the product arithmetic its FORWARD ran under (ctx.arith), not under
    whatever the host default is when .backward() happens to be called: a forward inside `with hb.arith("f32")` followed by
    a backward outside the block would otherwise mix arithmetics (and the fused-dW_hh decision with them)."""
    import functools

    @functools.wraps(bwd)
    def wrapped(ctx, *grads):
        with hb.arith(ctx.arith):
            return bwd(ctx, *grads)
    return wrapped


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, relu, drop):
        ctx.arith = hb.current_arith()
        x2 = x.reshape(-1, x.shape[-1])
        seeded = isinstance(drop, hb.SeededMask)
        if seeded:                                   # relu -> dropout in the product's own epilogue pass; the mask is
            assert relu and (x2.shape[0] * weight.shape[0]) % 4 == 0      # regenerated in the backward
        # (inside a train step the output is a pre-zeroed slice of the step's arena: a product the library splits over K
        # then needs no zero pass of its own)
        out = _ARENA.take((x2.shape[0], weight.shape[0]), x2.device) if _LINEAR_ARENA else None
        y = hb.gemm(x2, weight, trans_b=True, bias=bias, relu=relu, drop=drop if seeded else None, out=out,
                    out_zeroed=out is not None)
        ctx.save_for_backward(x2, weight, y if relu else None)
        ctx.side_mask = 0
        if _SIDE.mask_hint and x2.shape[0] >= 512 and ctx.needs_input_grad[1]:      # (grad mode is off inside forward: ask the node)
            ctx.side_mask = _SIDE.mask_hint
            _SIDE.uses[weight.data_ptr()] = _SIDE.uses.get(weight.data_ptr(), 0) + 1
        ctx.relu = relu
        ctx.drop = (drop.seed, drop.p) if seeded else None
        ctx.has_bias = bias is not None
        ctx.in_shape = x.shape
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    @_with_saved_arith
    def backward(ctx, dy):
        x2, weight, y = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        if ctx.relu and y.numel() % 4 == 0:
            # one pass: relu gradient (and the dropout mask: y is the dropped-out output, y > 0 <=> kept and active)
            # (its column sums - the bias gradient - from the same pass were tried: the atomics of a column meet in one L2
            # line, 32 us against 8 + 5 for the two launches at the layer-0 projection)
            seed, p = ctx.drop if ctx.drop is not None else (0, 0.0)
            dy2 = hb.relu_dropout_bwd(dy2, y, seed, p)
        elif ctx.relu:
            dy2 = dy2 * (y > 0).to(dy2.dtype)
        dx = hb.gemm(dy2, weight).view(ctx.in_shape) if ctx.needs_input_grad[0] else None
        # the weight gradient beside the recurrence of the layer below (a small batch; _SideStream) - unless this weight has
        # a second gradient on its way (the two model passes of the semi-supervised step: autograd would ADD the two on the
        # main stream while the side stream still writes them) or nothing follows this node in the pass
        if ctx.side_mask and ctx.needs_input_grad[0] and _SIDE.uses.get(weight.data_ptr(), 0) == 1 and _SIDE.usable(ctx.side_mask):
            dw = zeros_acc((dy2.shape[1], x2.shape[1]), dy2.device)
            queue = zeros_acc((1,), dy2.device)
            # (the closure writes through an ALIAS of dw - a tensor object of its own over the same memory: AccumulateGrad
            # takes a gradient over as .grad only while nobody else holds the tensor object, and CLONES it otherwise - at once,
            # on the main stream, i.e. the zeros)
            _SIDE.defer(dy2.device, _side_product(dy2, x2, dw.detach(), queue, ctx.side_mask), (dy2, x2, queue))
        else:
            dw = _gemm_acc(dy2, x2, trans_a=True, shape=(dy2.shape[1], x2.shape[1]))
        db = _colsum_acc(dy2) if ctx.has_bias else None
        return dx, dw, db, None, None


def linear(x, weight, bias=None, relu=False, drop=None):
    """nn.Linear (+ optional fused ReLU, + optional seeded dropout after it) on the f32 MFMA GEMM (model.py:93-95,144)."""
    return _Linear.apply(x, weight, bias, relu, drop)


# --------------------------------------------------------------------------------------
def _row_capacity(n):
    """Rows a pooled workspace is allocated for: n rounded up to a quarter of its power of two (<= 25 % over), so that the
    row counts of an epoch's batches (every batch has its own sum of lengths) share a handful of workspaces."""
    n = max(int(n), 1)
    q = max(256, 1 << max(n.bit_length() - 3, 0))
    return (n + q - 1) // q * q


def _lstm_workspace(rows, nbatch, H, ndir, dev, with_bwd):
    """Buffers of one LSTM layer for up to `rows` (time, batch) rows and `nbatch` batch rows; _lstm_views cuts the views of a
    call's actual shape out of them."""
    f32 = dict(device=dev, dtype=torch.float32)
    # (zeros, and one row more than asked for: the packed-row dW_hh product reads ONE row behind the matrix on each side -
    # against a zero padding row of the other operand - and that row must hold finite numbers: _LstmLayer.backward zeroes it
    # when a longer batch has used the workspace before)
    alloc = torch.zeros if with_bwd else torch.empty
    ws = dict(gates_buf=alloc(rows + 1, ndir, 4 * H, **f32), y_buf=alloc(rows + 1, ndir * H, **f32),
              c_buf=torch.empty(rows, ndir * H, **f32))
    if with_bwd:
        # the three accumulators the backward starts from zero share one buffer: one fill instead of three
        n1, n2, n3 = nbatch * ndir * H, ndir * 4 * H * H, ndir * 4 * H
        zbuf = torch.empty(n1 + n2 + n3, **f32)
        ws.update(w_hhT=torch.empty(ndir, H, 4 * H, **f32), dy_buf=torch.empty(rows, ndir * H, **f32), zbuf=zbuf,
                  dcarry=zbuf[:n1].view(nbatch, ndir * H), dw_hh=zbuf[n1:n1 + n2].view(ndir, 4 * H, H),
                  db=zbuf[n1 + n2:].view(ndir * 4 * H))
    return ws


def _lstm_views(ws, T, B, H, ndir):
    n = T * B
    ws["gates"] = ws["gates_buf"][:n].view(T, B, ndir, 4 * H)
    ws["y"] = ws["y_buf"][:n].view(T, B, ndir * H)
    ws["c"] = ws["c_buf"][:n].view(T, B, ndir * H)
    if "dy_buf" in ws:
        ws["dy"] = ws["dy_buf"][:n].view(T, B, ndir * H)
    return ws


class _LstmPack(torch.autograd.Function):
    """torch layout of nn.LSTM's parameters (per direction w_ih [4H,I], w_hh [4H,H], b_ih, b_hh; gate-major rows, model.py:
    67-68) -> the kernels' gate-interleaved layout, for ALL layers of a stack in one launch; the backward takes the layers'
    gradients in that layout back in one launch.  Outputs per layer: w_ih_cat [ndir*4H, I], w_hh_il [ndir, 4H, H], bias
    [ndir*4H] (= b_ih + b_hh)."""

    @staticmethod
    def forward(ctx, ndir, nlayers, *params):
        per = 4 * ndir
        layers = [params[per * j:per * (j + 1)] for j in range(nlayers)]
        ctx.ndir, ctx.dims = ndir, [(lp[1].shape[1], lp[0].shape[1]) for lp in layers]
        ctx.set_materialize_grads(False)
        outs = hb.lstm_pack_multi(layers, ndir)
        return tuple(t for out in outs for t in out)

    @staticmethod
    def backward(ctx, *grads):
        _SIDE.join_now()                   # the layers' dW_ih / dW_hh may still be on their way on the side stream
        n = len(ctx.dims)
        per_layer = [grads[3 * j:3 * j + 3] for j in range(n)]
        for j, g in enumerate(per_layer):
            if any(t is None for t in g) and not all(t is None for t in g):     # (never on the product's paths)
                H, I = ctx.dims[j]
                ref = next(t for t in g if t is not None)
                shp = ((ctx.ndir * 4 * H, I), (ctx.ndir, 4 * H, H), (ctx.ndir * 4 * H,))
                per_layer[j] = [t if t is not None else torch.zeros(shp[k], device=ref.device) for k, t in enumerate(g)]
        outs = hb.lstm_unpack_multi(per_layer, ctx.dims, ctx.ndir)
        flat = []
        for j in range(n):
            flat += outs[j] if outs[j] is not None else [None] * (4 * ctx.ndir)
        return (None, None) + tuple(flat)


def lstm_pack(layers, ndir):
    """layers: per layer the flat parameter list of its directions (model._LstmWeights.direction_params) -> per layer
    (w_ih_cat, w_hh_il, bias) for lstm_layer(packed=...)."""
    flat = [p for lp in layers for p in lp]
    out = _LstmPack.apply(ndir, len(layers), *flat)
    return [out[3 * j:3 * j + 3] for j in range(len(layers))]


class _LstmLayer(torch.autograd.Function):
    """One (bi)directional LSTM layer over a padded time-major batch (model.py:79-81).
    w_ih [ndir*4H, I], w_hh [ndir, 4H, H], bias [ndir*4H]: the gate-interleaved parameters (_LstmPack); their gradients
    leave in the same layout.
    rows (hb.LayerRows or None): the packed-row layout - x is then the row matrix [R, 1, I] (every product of this layer is
    a GEMM over the R rows as if it were a time-major batch of one; only the recurrence kernels know about utterances)."""

    @staticmethod
    def forward(ctx, x, lens, ndir, pooled, rows, w_ih, w_hh, bias):
        ctx.arith = hb.current_arith()
        T, B, I = x.shape
        H = w_hh.shape[2]
        dev = x.device
        x2 = x.reshape(T * B, I)
        nbatch = rows.B if rows is not None else B
        if pooled:
            cap = _row_capacity(T * B)
            lease = _POOL.acquire(("lstm", dev.index, cap, nbatch, H, ndir),
                                  lambda: _lstm_workspace(cap, nbatch, H, ndir, dev, True))
            ws = lease.ws
        else:
            lease, ws = None, _lstm_workspace(T * B, nbatch, H, ndir, dev, False)
        _lstm_views(ws, T, B, H, ndir)
        ws["rows_written"] = max(ws.get("rows_written", 0), T * B)      # rows of gates_buf / y_buf that may hold stale values
        ctx.side_mask = _SIDE.mask_hint = _SIDE.mask_for(nbatch) if pooled else 0
        ws["lens"] = lens                      # int32 device tensor, kept for the backward (no copy)
        hb.gemm(x2, w_ih, trans_b=True, bias=bias, out=ws["gates"].view(T * B, ndir * 4 * H))
        hb.lstm_seq_fwd(ws["gates"], w_hh, ws["lens"], ws["y"], ws["c"], use_graphs=pooled, rows=rows)
        ctx.save_for_backward(x2, w_ih, w_hh)
        ctx.lease = lease
        ctx.rows = rows
        ctx.dims = (T, B, I, H, ndir)
        return ws["y"].detach()      # fresh tensor object aliasing the workspace (no stale autograd metadata)

    @staticmethod
    @_with_saved_arith
    def backward(ctx, dy):
        x2, w_ih, w_hh = ctx.saved_tensors
        T, B, I, H, ndir = ctx.dims
        lease = ctx.lease
        assert lease is not None and lease.ws is not None, "lstm_layer backward needs the leased workspace " \
            "(autograd was off in forward, or backward ran twice)"
        ws = _lstm_views(lease.ws, T, B, H, ndir)      # (the views of THIS call's shape: the buffers are shared by capacity)
        dev = dy.device
        dyc = dy if dy.is_contiguous() else ws["dy"].copy_(dy)
        zb = _ARENA.take((ws["zbuf"].numel(),), dev)
        if zb is None:
            ws["zbuf"].zero_()                 # dcarry, dw_hh, db
        else:                                  # ... or their places in the step's arena (zeroed with everything else)
            nb_, n2 = ws["dcarry"].numel(), ws["dw_hh"].numel()
            ws = dict(ws, dcarry=zb[:nb_].view_as(ws["dcarry"]), dw_hh=zb[nb_:nb_ + n2].view_as(ws["dw_hh"]),
                      db=zb[nb_ + n2:].view_as(ws["db"]))
        gates, y = ws["gates"], ws["y"]
        _SIDE.flush()                          # weight-gradient products of the layers above: beside THIS layer's recurrence
        # the transposed recurrent weights are only formed if the kernel that reads the forward layout does not apply
        fused_dw, fused_db = hb.lstm_seq_bwd(gates, lambda: ws["w_hhT"].copy_(w_hh.transpose(1, 2)), ws["lens"], dyc,
                                             ws["c"], ws["dcarry"], y=y, dw_hh=ws["dw_hh"], db=ws["db"],
                                             w_hh=w_hh, rows=ctx.rows)                    # gates <- dG in place
        dG = gates.view(T * B, ndir * 4 * H)
        # a small batch: this layer's weight gradients go beside the recurrence of the layer BELOW (_SideStream) - not those of
        # the bottom layer (no input gradient: nothing follows it in the pass, and the main stream's kernels have the whole chip)
        side, side_dw_hh = False, False
        if ctx.side_mask and ctx.needs_input_grad[0] and _SIDE.usable(ctx.side_mask):
            side = True
            dw_ih = zeros_acc((ndir * 4 * H, I), dev)
            queue = zeros_acc((2,), dev)
            # (without a step arena the accumulators live in the leased workspace and are copied out below, on the main
            # stream: the side product then needs a tensor of its own)
            dw_hh_side = ws["dw_hh"] if zb is not None else torch.zeros_like(ws["dw_hh"])
            if ctx.rows is not None and not fused_dw and T > 1 and ws.get("rows_written", 0) > T * B:
                ws["gates_buf"][T * B].zero_()             # (row R of a workspace a longer batch has used: see below)
                ws["y_buf"][T * B].zero_()
        dx = hb.gemm(dG, w_ih).view(T, B, I) if ctx.needs_input_grad[0] else None
        if side:
            mask, with_hh, code = ctx.side_mask, (not fused_dw and T > 1), hb.current_arith()
            kk = T * B if ctx.rows is not None else (T - 1) * B

            def launch():
                # (dG and y live in the layer's pooled workspace: no forward pass leases it again before the pass has ended)
                hb.gemm_side(dG, x2, dw_ih, queue[0:1], mask, trans_a=True, arith=code)
                if with_hh:
                    ldg, ldy = ndir * 4 * H, ndir * H
                    hb.gemm_side_batched(dG, y, dw_hh_side, queue[1:2], mask, True, False, 4 * H, H, kk, ldg, ldy, H, ndir,
                                         4 * H - B * ldg, B * ldy + H, 4 * H * H, arith=code, a_off=B * ldg, b_off=0)
            _SIDE.defer(dev, launch, (x2, queue))
            side_dw_hh = with_hh
            fused_dw = True                                # (done: skip the main-stream product below)
        else:
            dw_ih = _gemm_acc(dG, x2, trans_a=True, shape=(ndir * 4 * H, I))      # [ndir*4H, I]
        db = ws["db"] if fused_db else _colsum_acc(dG)     # the persistent kernels sum the bias gradient themselves
        if not fused_dw and T > 1:
            # dW_hh[d] = sum_t dG_t[d]^T h_prev(t), h_prev = y[t-1] (d = 0) or y[t+1] (reverse direction): ONE batched GEMM
            # over the directions, K = (T-1)*B, accumulated into the zeroed dw_hh (no zero pass of its own).
            #   d = 0: A = dG[B:, 0:4H],        B = y[:(T-1)B, 0:H]
            #   d = 1: A = dG[:(T-1)B, 4H:8H],  B = y[B:, H:2H]          -> batch strides relative to d = 0 (may be negative)
            ldg, ldy = ndir * 4 * H, ndir * H
            # packed rows: K = R instead of R - 1 (a multiple of 4: the GEMM's fast kernels want that).  The one extra pair
            # is (row R of dG, the last row of y) resp. (the last row of dG, row R of y): the last row of the matrix is a
            # padding row - zero in both - and row R exists in the workspace and holds finite numbers (_lstm_workspace)
            kk = T * B if ctx.rows is not None else (T - 1) * B
            if ctx.rows is not None and ws.get("rows_written", 0) > T * B:
                # the workspace is shared by capacity: a LONGER batch left its own values in row R - finite ones normally,
                # but NaN when that batch's launch aborted (the kernels poison their outputs), and 0 * NaN is NaN
                ws["gates_buf"][T * B].zero_()
                ws["y_buf"][T * B].zero_()
            hb.gemm_batched(dG, y, ws["dw_hh"], True, False, 4 * H, H, kk, ldg, ldy, H, ndir,
                            4 * H - B * ldg, B * ldy + H, 4 * H * H, accumulate=True, a_off=B * ldg, b_off=0)
        # the gradients stay gate-interleaved (_LstmPack.backward converts every layer's in one launch); what lives in the
        # leased workspace is copied out of it, slices of the step's arena outlive the lease
        dw_hh, db = (ws["dw_hh"], db) if zb is not None else (ws["dw_hh"].clone(), db.clone() if fused_db else db)
        if side_dw_hh:
            dw_hh = dw_hh_side
        lease.release()
        return dx, None, None, None, None, dw_ih, dw_hh, db


def lstm_layer(x, lens, params, ndir, rows=None, packed=None):
    """x [T,B,I] time-major contiguous, lens int32 device [B] -> y [T,B,ndir*H].
    rows (hb.LayerRows): packed rows - x [R, I] -> y [R, ndir*H], lens = rows.lens.
    packed: this layer's (w_ih_cat, w_hh_il, bias) from lstm_pack (a stack converts all its layers in one launch); None:
    converted here from the torch-layout `params`."""
    if packed is None:
        packed = lstm_pack([params], ndir)[0]
    w_ih, w_hh, bias = packed
    pooled = torch.is_grad_enabled() and (x.requires_grad or w_hh.requires_grad)
    if rows is not None:
        return _LstmLayer.apply(x.contiguous().view(rows.R, 1, -1), rows.lens, ndir, pooled, rows, w_ih, w_hh, bias).view(rows.R, -1)
    return _LstmLayer.apply(x.contiguous(), lens, ndir, pooled, None, w_ih, w_hh, bias)


# --------------------------------------------------------------------------------------
class _Pyramid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask, rep):
        T, B, C = x.shape
        out = torch.empty((T + 1) // 2, B, 2 * C, device=x.device, dtype=torch.float32)
        hb.pyramid_fwd(x.contiguous(), mask, out)
        if rep is not None:                        # packed rows: the replicate-padded frame of the longest utterances
            out[rep, 0, C:] = out[rep, 0, :C]
        ctx.mask = mask                            # tensor, hb.SeededMask (regenerated in the backward) or None
        ctx.rep = rep
        ctx.shape = (T, B, C)
        return out

    @staticmethod
    def backward(ctx, dout):
        T, B, C = ctx.shape
        din = torch.empty(ctx.shape, device=dout.device, dtype=torch.float32)
        dout = dout.contiguous()
        if ctx.rep is not None:                    # the replica's gradient belongs to the frame it copies; the padding row
            dout = dout.clone()                    # whose place it took gets none
            dout[ctx.rep, 0, :C] += dout[ctx.rep, 0, C:]
            dout[ctx.rep, 0, C:] = 0.0
        hb.pyramid_bwd(dout, ctx.mask, din)
        return din, None, None


def pyramid_concat(x, mask=None, rep_rows=None):
    """[T,B,C] -> [ceil(T/2),B,2C] (model.py:85-92); `mask` = pre-scaled dropout mask or None.
    Packed rows: x [R, 1, C] (R even) -> [R / 2, 1, 2C]; rep_rows = long tensor of the output rows whose second half is the
    reference's replicate-padded frame (hb.RowLayout.replicated_rows), or None."""
    return _Pyramid.apply(x, mask, rep_rows)


class _RowsUnpack(torch.autograd.Function):
    """Packed encoder output [R, C] -> the padded batch [B, T, C] the decoder reads (model.py:109-112).  The frames behind an
    utterance are what the reference's last projection makes of a zero frame, dropout(relu(bias)): fill [C] * mask."""

    @staticmethod
    def forward(ctx, packed, fill, rows, T, mask, fill_relu):
        ctx.rows, ctx.mask, ctx.C = rows, mask, packed.shape[1]
        ctx.want_fill = fill is not None and fill.requires_grad
        fc = fill.contiguous() if fill is not None else None
        ctx.relu_of = fc if (fill_relu and ctx.want_fill) else None
        return hb.rows_unpack_fwd(packed.contiguous(), rows, T, fc, mask, fill_relu=fill_relu)

    @staticmethod
    def backward(ctx, dout):
        acc = _ARENA.take((ctx.C,), dout.device) if ctx.want_fill else None
        drows, dfill = hb.rows_unpack_bwd(dout.contiguous(), ctx.rows, ctx.C, ctx.mask, ctx.want_fill, relu_of=ctx.relu_of,
                                          dfill=acc)
        return drows, dfill, None, None, None, None


def rows_unpack(packed, rows, T, fill=None, mask=None, fill_relu=False):
    """fill_relu: the padded frames hold relu(fill) * mask - `fill` is the last projection's bias as it is (no relu launch in
    front, none of its backward behind)."""
    return _RowsUnpack.apply(packed, fill, rows, T, mask, fill_relu)


# --------------------------------------------------------------------------------------
def _p(t, off=0):
    """device pointer of tensor t advanced by `off` float elements (None stays NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr() + 4 * int(off))


def _dec_fwd_struct(d, b0, nb):
    """asr_dec_fwd_t for rows [b0, b0+nb) of the decoder buffers in dict d (pointers pre-offset, B = stride)."""
    Tp, A, D, O, E, C, KX = d["Tp"], d["A"], d["D"], d["O"], d["E"], d["C"], d["KX"]
    return hb.DecFwd(
        B=d["B"], nb=nb, Tp=Tp, A=A, D=D, O=O, E=E, C=C, K=d["K"], L=d["L"], scaling=d["scaling"],
        P=_p(d["P"], b0 * Tp * A), Q=_p(d["Q"], b0 * Tp * O), bo=_p(d["bo"]), wcat=_p(d["wcat"]), bcat=_p(d["bcat"]),
        wdec=_p(d["wdec"]), convw=_p(d["convw"]), watt=_p(d["watt"]), wattT=_p(d["wattT"]), gvec=_p(d["gvec"]), w0=_p(d["w0"], b0 * Tp),
        xmask=_p(d["xmask"], b0 * (O + E)), X=_p(d["X"], b0 * KX), Xd=_p(d["Xd"], b0 * KX),
        gates=_p(d["gates"], b0 * 4 * D), cstate=_p(d["cstate"], b0 * D), Dproj=_p(d["Dproj"], b0 * A),
        fconv=_p(d["fconv"], b0 * C * Tp), S=_p(d["S"], b0 * Tp * A), energy=_p(d["energy"], b0 * Tp),
        ws=_p(d["ws"], b0 * Tp))


def _dec_bwd_struct(d, w, b0, nb):
    Tp, A, D, C, KX = d["Tp"], d["A"], d["D"], d["C"], d["KX"]
    taps = 2 * d["K"] + 1
    return hb.DecBwd(
        f=_dec_fwd_struct(d, b0, nb), wcatT=_p(w["wcatT"]), wdecT=_p(w["wdecT"]), dws=_p(w["dws"], b0 * Tp),
        G=_p(w["G"], b0 * KX), dwext=_p(w["dwext"], b0 * Tp), dwraw=_p(w["dwraw"], b0 * Tp),
        dfpart=_p(w["dfpart"], b0 * C * Tp), dP=_p(w["dP"], b0 * Tp * A), dgates=_p(w["dgates"], b0 * 4 * D),
        dD=_p(w["dD"], b0 * A), dcell=_p(w["dcell"], b0 * D), dgvec_part=_p(w["dgvec_part"], b0 * A),
        dwatt_part=_p(w["dwatt_part"], b0 * A * C), dconv_part=_p(w["dconv_part"], b0 * C * taps))


def _dec_workspace(B, Tp, A, D, O, E, C, K, L, drop, dev, with_bwd):
    f32 = dict(device=dev, dtype=torch.float32)
    KX = D + O + E
    ws = dict(
        P=torch.empty(B, Tp, A, **f32), Q=torch.empty(B, Tp, O, **f32), wcat=torch.empty(4 * D, KX, **f32),
        bcat=torch.empty(4 * D, **f32), convw=torch.empty(C, 2 * K + 1, **f32), gvec=torch.empty(A, **f32),
        wattT=torch.empty(C, A, **f32),
        w0=torch.empty(B, Tp, **f32), X=torch.empty(L + 1, B, KX, **f32),
        Xd=torch.empty(L + 1, B, KX, **f32) if drop else None,
        xmask=torch.empty(L, B, O + E, **f32) if drop else None,
        gates=torch.empty(L, B, 4 * D, **f32), cstate=torch.empty(L, B, D, **f32), Dproj=torch.empty(L, B, A, **f32),
        fconv=torch.empty(L, B, C, Tp, **f32), S=torch.empty(L, B, Tp, A, **f32), energy=torch.empty(L, B, Tp, **f32),
        ws=torch.empty(L, B, Tp, **f32))
    if with_bwd:
        ntile = (A + 63) // 64
        # everything the backward accumulates into lives in ONE buffer (16-byte aligned slices): one fill per step
        shapes = dict(G=(L + 1, B, KX), dwext=(C, B, Tp), dP=(B, Tp, A), dcell=(B, D), dgvec_part=(B, A),
                      dwatt_part=(B, A, C), dconv_part=(B, C, 2 * K + 1))
        sizes = {k: (int(np.prod(v)) + 3) // 4 * 4 for k, v in shapes.items()}
        zbuf = torch.empty(sum(sizes.values()), **f32)
        off = 0
        for k, shp in shapes.items():
            ws[k] = zbuf[off:off + int(np.prod(shp))].view(*shp)
            off += sizes[k]
        ws.update(
            zbuf=zbuf, wcatT=torch.empty(KX, 4 * D, **f32), wdecT=torch.empty(D, A, **f32),
            dwraw=torch.empty(B, Tp, **f32), dfpart=torch.empty(ntile, B, C, Tp, **f32),
            dgates=torch.empty(L, B, 4 * D, **f32), dD=torch.empty(L, B, A, **f32),
            dws=torch.empty(L, B, Tp, **f32), Mf=torch.empty(L, B, C, Tp, **f32))
    return ws


class _DecoderSeq(torch.autograd.Function):
    """All decoder steps (Decoder.forward loop, model.py:324-351) as one graph node.

    inputs : P [B,Tp,A], Q [B,Tp,O] (= enc_h W_o^T, no bias), emb_w [V,E], w_ih [4D,E+O], w_hh [4D,D],
             b_ih, b_hh, wdec [A,D], convw [C,1,1,2K+1], watt [A,C], gvec [1,A], bo [O], w_out [V,D+O],
             b_out [V], w0 [B,Tp]
    opts   : dict(L, tokens [B,L] long or None, tf_flags list[bool] or None, smooth, smooth_scaling,
                  sample, scaling (attention temperature), xmask [L,B,O+E] or None, pooled)
    returns: logits [L,B,V], ws [L,B,Tp], prediction [L,B] (long)
    """

    @staticmethod
    def forward(ctx, P, Q, emb_w, w_ih, w_hh, b_ih, b_hh, wdec, convw, watt, gvec, bo, w_out, b_out, w0, opts):
        ctx.arith = hb.current_arith()
        dev = P.device
        B, Tp, A = P.shape
        O = Q.shape[2]
        D = w_hh.shape[1]
        E = emb_w.shape[1]
        V = w_out.shape[0]
        C = convw.shape[0]
        K = (convw.shape[-1] - 1) // 2
        L = int(opts["L"])
        KX = D + O + E
        f32 = dict(device=dev, dtype=torch.float32)
        xmask_in = opts.get("xmask")
        drop = xmask_in is not None
        pooled = bool(opts.get("pooled", False))
        if pooled:
            lease = _POOL.acquire(("dec", dev.index, B, Tp, A, D, O, E, C, K, L, drop),
                                  lambda: _dec_workspace(B, Tp, A, D, O, E, C, K, L, drop, dev, True))
            ws = lease.ws
        else:
            lease, ws = None, _dec_workspace(B, Tp, A, D, O, E, C, K, L, drop, dev, False)
        # [4D, KX] gate-interleaved rows + the transposed images the per-step forward (wattT) and the backward (wcatT, wdecT)
        # read, one launch
        hb.dec_pack(w_ih, w_hh, b_ih, b_hh, wdec, watt, D, O, E, A, C, ws["wcat"], ws["bcat"], ws.get("wcatT"),
                    ws.get("wdecT"), ws["wattT"])
        # inputs used as they are (no staging copies); the dict keeps them alive until the backward has run
        ws["convw"] = convw.reshape(C, 2 * K + 1).contiguous()
        ws["gvec"] = gvec.reshape(A).contiguous()
        ws["P"], ws["Q"], ws["w0"] = P.contiguous(), Q.contiguous(), w0.contiguous()
        if drop:
            ws["xmask"] = xmask_in.contiguous()
        X, Xd, xmask = ws["X"], ws["Xd"], ws["xmask"]
        wdec_c, watt_c, bo_c, w_out_c = wdec.contiguous(), watt.contiguous(), bo.contiguous(), w_out.contiguous()
        d = dict(B=B, Tp=Tp, A=A, D=D, O=O, E=E, C=C, K=K, L=L, KX=KX, scaling=float(opts.get("scaling", 2.0)),
                 bo=bo_c, wdec=wdec_c, watt=watt_c)
        d.update({k: ws[k] for k in ("P", "Q", "wcat", "bcat", "convw", "gvec", "wattT", "w0", "xmask", "X", "Xd", "gates",
                                     "cstate", "Dproj", "fconv", "S", "energy", "ws")})
        lib = hb.load()
        tokens = opts.get("tokens")
        tf_flags = opts.get("tf_flags")
        smooth = bool(opts.get("smooth", False))
        sample = bool(opts.get("sample", False))
        all_teacher = tokens is not None and (tf_flags is None or all(tf_flags)) and not sample
        fused_prep = (all_teacher and (D + O) % 4 == 0 and E % 4 == 0 and (O + E) % 4 == 0 and X.shape[0] == L + 1 and
                      X.is_contiguous() and tokens.stride(1) == 1)
        probs_saved = []
        if fused_prep:                           # zero fills, embedding gather, input dropout, fed: one launch
            fed = torch.empty(L, B, dtype=torch.long, device=dev)
            hb.dec_prepare(tokens, emb_w.contiguous(), xmask if drop else None, X, Xd if drop else None, fed, L, B, D, O, E)
        else:
            fed = torch.zeros(L, B, dtype=torch.long, device=dev)   # token whose embedding fed step s (-1: smooth)
            X.zero_()
            if drop:
                Xd.zero_()
        if all_teacher:
            if not fused_prep:
                fed.copy_(tokens.t())
                X[:L, :, D + O:] = emb_w[fed]
                if drop:
                    Xd[:L, :, D + O:] = X[:L, :, D + O:] * xmask[:, :, O:]
            groups = hb.row_groups(B)
            done = False
            if hb.USE_PERSIST_DEC and len(groups) == 1:          # one launch for the whole sequence
                fg = _dec_fwd_struct(d, 0, B)
                xch, ctrl = hb.persist_scratch(dev)
                entry = lib.asr_dec_seq_fwd_persist_fault if hb.DEC_FAULT[0] else lib.asr_dec_seq_fwd_persist
                rc = entry(ctypes.byref(fg), ctypes.c_void_p(xch.data_ptr()), ctypes.c_void_p(ctrl.data_ptr()), hb.stream())
                if rc == 0:
                    done = True
                elif rc != -2:                                  # -2: shape/device not covered by the fast path
                    hb.check(rc, "asr_dec_seq_fwd_persist")
            hb.count_path("dec_fwd", done, "D=%d A=%d O=%d E=%d Tp=%d B=%d" % (D, A, O, E, Tp, B))
            if not done:
                gh = [hb.graphs_for(i) if pooled else None for i in range(len(groups))]

                def run(gi, grp, st):
                    fg = _dec_fwd_struct(d, grp[0], grp[1])
                    hb.check(lib.asr_dec_seq_fwd(ctypes.byref(fg), 0, L, gh[gi], st), "asr_dec_seq_fwd")

                hb.run_grouped(groups, run)
            logits = hb.gemm(X[1:].view(L * B, KX)[:, :D + O], w_out_c, trans_b=True, bias=b_out).view(L, B, V)
            # (skip_pred: the caller takes the argmax from the loss kernel that reads the logits anyway - label_logprob)
            pred = None if opts.get("skip_pred") else logits.argmax(-1)
        else:
            fs = _dec_fwd_struct(d, 0, B)
            logits = torch.empty(L, B, V, **f32)
            pred = torch.empty(L, B, dtype=torch.long, device=dev)
            done = False
            if hb.USE_FEEDBACK_KERNEL and not sample and V <= 128:
                # free-running steps: per step the decoder chain, then ONE kernel for logits + argmax + the next
                # step's embedding input (teacher / predicted token, or the smooth embedding softmax(k*logit) @ E)
                emb_c = emb_w.contiguous()
                if smooth and tokens is None:
                    probs_saved = torch.empty(max(L - 1, 1), B, V, **f32)
                tok_c = tokens.contiguous() if tokens is not None else None
                fed[0] = tok_c[:, 0] if tok_c is not None else opts["bos"]
                X[0, :, D + O:] = emb_c[fed[0]]
                if drop:
                    Xd[0, :, D + O:] = X[0, :, D + O:] * xmask[0, :, O:]
                if hb.USE_PERSIST_DEC and (tok_c is None or not smooth) and V <= 64 and len(hb.row_groups(B)) == 1:
                    # the whole sequence in one launch, the feedback computed in the kernel: no teacher tokens at all, or
                    # scheduled sampling (the host's per-step draws go along as a byte per step)
                    # decoding without autograd: a group of 4 utterances stops once all of them have emitted <EOS>; the
                    # outputs of the steps that are not run read <EOS> / zero logits / zero attention weights
                    eos = int(opts.get("eos", -1))
                    stop = hb.DECODE_EARLY_STOP and eos >= 0 and not torch.is_grad_enabled() and tok_c is None
                    if stop:
                        pred.fill_(eos)
                        logits.zero_()
                        ws["ws"].zero_()
                    tf_dev = None
                    if tok_c is not None:
                        tf_dev = torch.tensor([1 if (tf_flags is None or tf_flags[i]) else 0 for i in range(L)],
                                              dtype=torch.uint8).to(dev, non_blocking=True)
                    fb = hb.DecFeedback(
                        tokens=ctypes.c_void_p(tok_c.data_ptr()) if tok_c is not None else None,
                        ld_tokens=int(tok_c.stride(0)) if tok_c is not None else 0,
                        teacher=ctypes.c_void_p(tf_dev.data_ptr()) if tf_dev is not None else None,
                        mode=2 if smooth else 1, V=V, eos=eos if stop else -1,
                        scaling=float(opts.get("smooth_scaling", 1.0)), w_out=_p(w_out_c),
                        b_out=_p(b_out.contiguous()), emb=_p(emb_c), logits=_p(logits),
                        probs=_p(probs_saved) if smooth else None, pred=ctypes.c_void_p(pred.data_ptr()),
                        fed=ctypes.c_void_p(fed.data_ptr()))
                    xch, ctrl = hb.persist_scratch(dev)
                    rc = lib.asr_dec_seq_fwd_persist_free(ctypes.byref(fs), ctypes.byref(fb), ctypes.c_void_p(xch.data_ptr()),
                                                          ctypes.c_void_p(ctrl.data_ptr()), hb.stream())
                    if rc == 0:
                        done = True
                        if stop and L > 1:
                            # the last step's logits come from X[L], which a stopped group never wrote: rows that had
                            # already emitted <EOS> keep the pre-filled outputs
                            lg_last, pr_last = torch.empty(B, V, **f32), torch.empty(B, dtype=torch.long, device=dev)
                            hb.dec_feedback_fwd(X[L][:, :D + O], w_out_c, b_out, emb_c, lg_last, pr_last, hb.FEED_NONE)
                            live = pred[:L - 1].ne(eos).all(0)
                            pred[L - 1] = torch.where(live, pr_last, pred[L - 1])
                            logits[L - 1] = torch.where(live.unsqueeze(1), lg_last, logits[L - 1])
                        else:
                            hb.dec_feedback_fwd(X[L][:, :D + O], w_out_c, b_out, emb_c, logits[L - 1], pred[L - 1],
                                                hb.FEED_NONE)
                    elif rc != -2:
                        hb.check(rc, "asr_dec_seq_fwd_persist_free")
                hb.count_path("dec_free", done, "D=%d A=%d O=%d E=%d Tp=%d B=%d V=%d teacher=%s" % (
                    D, A, O, E, Tp, B, V, tok_c is not None))
                for s in (range(L) if not done else ()):
                    hb.check(lib.asr_dec_step_fwd(ctypes.byref(fs), s, hb.stream()), "asr_dec_step_fwd")
                    last = s == L - 1
                    if last:
                        mode = hb.FEED_NONE
                    elif tok_c is not None:
                        mode = hb.FEED_TEACHER if (tf_flags is None or tf_flags[s + 1]) else hb.FEED_PREDICTED
                    else:
                        mode = hb.FEED_SMOOTH if smooth else hb.FEED_PREDICTED
                    hb.dec_feedback_fwd(
                        X[s + 1][:, :D + O], w_out_c, b_out, emb_c, logits[s], pred[s], mode, opts["smooth_scaling"],
                        tok=tok_c[:, s + 1] if mode == hb.FEED_TEACHER else None, fed=None if last else fed[s + 1],
                        probs=probs_saved[s] if mode == hb.FEED_SMOOTH else None,
                        x_emb_next=None if last else X[s + 1][:, D + O:],
                        xd_emb_next=Xd[s + 1][:, D + O:] if (drop and not last) else None,
                        mask=xmask[s + 1][:, O:] if (drop and not last) else None)
            elif not done:
                for s in range(L):
                    if s == 0:
                        tok = tokens[:, 0] if tokens is not None else torch.full((B,), opts["bos"], dtype=torch.long,
                                                                                 device=dev)
                        fed[0] = tok
                        X[0, :, D + O:] = emb_w[tok]
                    elif tokens is not None:
                        tok = tokens[:, s] if tf_flags[s] else pred[s - 1]
                        fed[s] = tok
                        X[s, :, D + O:] = emb_w[tok]
                    elif not smooth:
                        fed[s] = pred[s - 1]
                        X[s, :, D + O:] = emb_w[pred[s - 1]]
                    else:
                        pr = torch.softmax(logits[s - 1] * opts["smooth_scaling"], dim=-1)
                        probs_saved.append(pr)
                        fed[s] = -1
                        hb.gemm(pr, emb_w, out=X[s][:, D + O:])
                    if drop:
                        Xd[s, :, D + O:] = X[s, :, D + O:] * xmask[s, :, O:]
                    hb.check(lib.asr_dec_step_fwd(ctypes.byref(fs), s, hb.stream()), "asr_dec_step_fwd")
                    hb.gemm_skinny(X[s + 1][:, :D + O], w_out_c, bias=b_out, out=logits[s])
                    pred[s] = torch.distributions.Categorical(logits=logits[s]).sample() if sample \
                        else logits[s].argmax(-1)
        ctx.d = d
        ctx.lease = lease
        ctx.keep = (wdec_c, watt_c, bo_c, fed, probs_saved, w_out_c, emb_w)
        ctx.dims = (B, Tp, A, O, D, E, V, C, K, L, KX)
        ctx.smooth = smooth and tokens is None
        ctx.all_teacher = all_teacher
        # a scheduled-sampling sequence that ran in the persistent kernel takes the persistent backward as well: no gradient
        # flows through an argmax, the backward only needs what the forward saved (X, fed)
        ctx.free_persist = (not all_teacher) and bool(done) and not (smooth and tokens is None)
        ctx.smooth_scaling = float(opts.get("smooth_scaling", 1.0))
        if pred is not None:
            ctx.mark_non_differentiable(pred)
        ctx.set_materialize_grads(False)       # an unused `ws` output arrives as None instead of a zero tensor + copy
        # the attention weights: in a training step a view of the leased workspace (valid until the next forward of the same
        # shape; nothing on the training path keeps them), a copy otherwise
        return logits, (ws["ws"].detach() if pooled else ws["ws"].clone()), pred

    @staticmethod
    @_with_saved_arith
    def backward(ctx, dlogits, dws, _dpred):
        wdec, watt, bo, fed, probs_saved, w_out, emb_w = ctx.keep
        B, Tp, A, O, D, E, V, C, K, L, KX = ctx.dims
        lease, d = ctx.lease, ctx.d
        assert lease is not None and lease.ws is not None, "decoder_sequence backward needs the leased workspace"
        wk = lease.ws
        X, Xd = wk["X"], wk["Xd"]
        dev = X.device
        lib = hb.load()
        if dlogits is None:
            dlogits = torch.zeros(L, B, V, device=dev, dtype=torch.float32)
        dlog2 = dlogits.contiguous().view(L * B, V)
        zb = _ARENA.take((wk["zbuf"].numel(),), dev)
        if zb is None:
            wk["zbuf"].zero_()                 # G, dwext, dP, dcell, dgvec_part, dwatt_part, dconv_part
        else:                                  # ... or their places in the step's arena (zeroed with everything else)
            wk = dict(wk)
            for k_ in ("G", "dwext", "dP", "dcell", "dgvec_part", "dwatt_part", "dconv_part"):
                o_ = (wk[k_].data_ptr() - wk["zbuf"].data_ptr()) // 4
                wk[k_] = zb[o_:o_ + wk[k_].numel()].view_as(wk[k_])
        G = wk["G"]                            # (wcatT, wdecT: written by the forward's dec_pack)
        XO = X[1:].view(L * B, KX)[:, :D + O]
        hb.gemm(dlog2, w_out, out=G[1:].view(L * B, KX)[:, :D + O])
        dw_out = _gemm_acc(dlog2, XO, trans_a=True, shape=(V, D + O))
        db_out = _colsum_acc(dlog2)
        w = dict(wk)
        if dws is not None:
            wk["dws"].copy_(dws)
        else:
            w["dws"] = None
        demb_w = zeros_acc(tuple(emb_w.shape), dev)
        if not ctx.smooth:
            groups = hb.row_groups(B)
            done = False
            if hb.USE_PERSIST_DEC_BWD and (ctx.all_teacher or ctx.free_persist) and len(groups) == 1:
                bg = _dec_bwd_struct(d, w, 0, B)
                xch, ctrl = hb.persist_scratch(dev)
                rc = lib.asr_dec_seq_bwd_persist(ctypes.byref(bg), _p(wk["Mf"]), ctypes.c_void_p(xch.data_ptr()),
                                                 ctypes.c_void_p(ctrl.data_ptr()), hb.stream())
                if rc == 0:
                    done = True
                elif rc != -2:
                    hb.check(rc, "asr_dec_seq_bwd_persist")
            hb.count_path("dec_bwd", done, "D=%d A=%d O=%d E=%d Tp=%d B=%d teacher=%s" % (D, A, O, E, Tp, B, ctx.all_teacher))
            if not done:
                gh = [hb.graphs_for(i) for i in range(len(groups))]

                def run(gi, grp, st):
                    bg = _dec_bwd_struct(d, w, grp[0], grp[1])
                    hb.check(lib.asr_dec_seq_bwd(ctypes.byref(bg), 0, L, gh[gi], st), "asr_dec_seq_bwd")

                hb.run_grouped(groups, run)
        else:
            # smooth-embedding feedback (model.py:341): emb_s = softmax(logit_{s-1}*k) @ E couples step s to
            # the logits of step s-1, so the extra gradient is injected between the per-step kernels.
            bs = _dec_bwd_struct(d, w, 0, B)
            k = ctx.smooth_scaling
            fused = torch.is_tensor(probs_saved)
            done = False
            if fused and hb.USE_PERSIST_DEC_BWD and L > 1 and len(hb.row_groups(B)) == 1:
                # the whole free-running sequence in one launch: the feedback path (d(emb_s) -> logit_{s-1} -> [z, ctx]_{s-1})
                # is carried inside the persistent kernel (dec_persist.hip, template FB)
                emb_c, w_out_c = emb_w.contiguous(), w_out.contiguous()
                dlfb = torch.zeros(L, B, V, device=dev, dtype=torch.float32)
                fbs = hb.DecFeedbackBwd(V=V, scaling=float(k), w_out=_p(w_out_c), emb=_p(emb_c), probs=_p(probs_saved),
                                        dlfb=_p(dlfb))
                xch, ctrl = hb.persist_scratch(dev)
                rc = lib.asr_dec_seq_bwd_persist_free(ctypes.byref(bs), ctypes.byref(fbs), _p(wk["Mf"]),
                                                      ctypes.c_void_p(xch.data_ptr()), ctypes.c_void_p(ctrl.data_ptr()),
                                                      hb.stream())
                if rc == 0:
                    done = True
                    dtot = dlog2.view(L, B, V) + dlfb
                    dw_out = hb.gemm(dtot.view(L * B, V), XO, trans_a=True)
                    db_out = hb.colsum(dtot.view(L * B, V))
                    hb.gemm(probs_saved[:L - 1].view((L - 1) * B, V), G[1:L].view((L - 1) * B, KX)[:, D + O:], trans_a=True,
                            out=demb_w, accumulate=True, split_k=1)
                elif rc != -2:
                    hb.check(rc, "asr_dec_seq_bwd_persist_free")
            hb.count_path("dec_bwd", done, "free-running smooth: D=%d A=%d O=%d E=%d Tp=%d B=%d V=%d L=%d fused-feedback=%s" % (
                D, A, O, E, Tp, B, V, L, fused))
            if done:
                pass
            elif fused:
                # one kernel per step carries the embedding gradient back into logit_{s-1} and [z_{s-1}, c_{s-1}];
                # the weight gradients that depend on it are taken once over the whole sequence afterwards
                dtot = dlog2.clone().view(L, B, V)
                emb_c, w_out_c = emb_w.contiguous(), w_out.contiguous()
                for s in range(L - 1, -1, -1):
                    hb.check(lib.asr_dec_step_bwd(ctypes.byref(bs), s, hb.stream()), "asr_dec_step_bwd")
                    if s >= 1:
                        hb.dec_feedback_bwd(G[s][:, D + O:], G[s][:, :D + O], probs_saved[s - 1], emb_c, w_out_c, k,
                                            dtot[s - 1])
                dw_out = hb.gemm(dtot.view(L * B, V), XO, trans_a=True)
                db_out = hb.colsum(dtot.view(L * B, V))
                if L > 1:
                    hb.gemm(probs_saved[:L - 1].view((L - 1) * B, V), G[1:L].view((L - 1) * B, KX)[:, D + O:], trans_a=True,
                            out=demb_w, accumulate=True, split_k=1)
            for s in (range(L - 1, -1, -1) if not (fused or done) else ()):
                hb.check(lib.asr_dec_step_bwd(ctypes.byref(bs), s, hb.stream()), "asr_dec_step_bwd")
                if s >= 1:
                    demb = G[s][:, D + O:]
                    pr = probs_saved[s - 1]
                    hb.gemm(pr, demb, trans_a=True, out=demb_w, accumulate=True, split_k=1)
                    dp = hb.gemm(demb, emb_w, trans_b=True)
                    dl = (k * pr * (dp - (pr * dp).sum(-1, keepdim=True))).contiguous()
                    hb.gemm(dl, w_out, out=G[s][:, :D + O], accumulate=True, split_k=1)
                    hb.gemm(dl, X[s][:, :D + O], trans_a=True, out=dw_out, accumulate=True, split_k=1)
                    hb.colsum(dl, out=db_out, accumulate=True)
        # deferred weight gradients: one GEMM each over the whole sequence
        dg2 = wk["dgates"].view(L * B, 4 * D)
        Xin = X[:L] if Xd is None else Xd[:L]
        dwcat = _gemm_acc(dg2, Xin.reshape(L * B, KX), trans_a=True, shape=(4 * D, KX))     # [4D, KX] gate-interleaved rows
        dw_ih, dw_hh, dbias, dbias2 = hb.cell_unpack(dwcat, _colsum_acc(dg2), D, O, E)       # -> torch layout, one launch
        dwdec = _gemm_acc(wk["dD"].view(L * B, A), X[1:].view(L * B, KX)[:, :D], trans_a=True, shape=(A, D))
        dgvec, dwatt, dconvw = hb.colsum_parts([wk["dgvec_part"], wk["dwatt_part"], wk["dconv_part"]])   # sums over utterances
        dgvec, dconvw = dgvec.view(1, A), dconvw.view(C, 1, 1, 2 * K + 1)
        # dQ[b] = ws[:, b, :]^T dctx[:, b, :]   (batched over utterances)
        dQ = torch.empty(B, Tp, O, device=dev, dtype=torch.float32)
        dctx_base = G[1:]                               # [L, B, KX], ctx grad at columns D:D+O
        hb.gemm_batched(wk["ws"], dctx_base[:, :, D:], dQ, True, False, Tp, O, L, B * Tp, B * KX, O, B, Tp, KX,
                        Tp * O)
        dbo = _colsum_acc(dctx_base.view(L * B, KX)[:, D:D + O])
        # embedding gradient for token-fed steps: one launch over the embedding columns of G as they lie (fed = -1: a step
        # whose input was not a token)
        demb_all = G[:L, :, D + O:]
        nsteps = 1 if torch.is_tensor(probs_saved) else L     # smooth feedback: only step 0 was fed a token (<BOS>)
        if not hb.embedding_grad(fed[:nsteps].reshape(-1), G[:nsteps].view(nsteps * B, KX)[:, D + O:], demb_w):
            if torch.is_tensor(probs_saved):
                demb_w.index_add_(0, fed[0], demb_all[0])
            elif not probs_saved:    # every step was fed a token (decided on the host: no device sync here)
                demb_w.index_add_(0, fed.view(-1), demb_all.reshape(L * B, E))
            else:
                tokfed = fed >= 0
                demb_w.index_add_(0, fed[tokfed], demb_all[tokfed])
        dP = wk["dP"] if zb is not None else wk["dP"].clone()      # (an arena slice outlives the lease)
        lease.release()
        return (dP, dQ, demb_w, dw_ih, dw_hh, dbias, dbias2, dwdec, dconvw, dwatt, dgvec, dbo, dw_out, db_out,
                None, None)


class _LabelLogProb(torch.autograd.Function):
    """(1-ls) log_softmax(logits)[target] + ls sum_v labeldist_v log_softmax(logits)_v  (model.py:354-366) in one
    kernel each way.  logits [..., V] contiguous, index [...] long -> ([...], total, argmax): `total` (with_sum) is
    sum_scale times the sum of all outputs, accumulated by the same kernel - the training loss is that sum times a constant
    (solver.py:377; sum_scale = that constant makes `total` the loss itself), so neither a reduction kernel nor a multiply nor
    their backward follow; a gradient that arrives through `total` alone is one device scalar, broadcast by the backward
    kernel.  argmax (with_argmax): the row's argmax over V, long - the `prediction` output of model.py:346, read off the
    logits the kernel has in its registers anyway."""

    @staticmethod
    def forward(ctx, logits, index, labeldist, ls_weight, with_sum, sum_scale, with_argmax):
        lg = logits.contiguous()
        V = lg.shape[-1]
        rows = lg.numel() // V
        idx = index.contiguous()
        out = torch.empty(lg.shape[:-1], device=lg.device, dtype=torch.float32)
        total = zeros_acc((1,), lg.device).view(()) if with_sum else None
        amax = torch.empty(lg.shape[:-1], device=lg.device, dtype=torch.long) if with_argmax else None
        dist = labeldist.contiguous() if labeldist is not None else None
        hb.check(hb.load().asr_label_logprob_fwd(rows, V, hb.ptr(lg), V, ctypes.c_void_p(idx.data_ptr()), hb.ptr(dist),
                                                 float(ls_weight), hb.ptr(out), hb.ptr(total), float(sum_scale),
                                                 None if amax is None else ctypes.c_void_p(amax.data_ptr()), hb.stream()),
                 "asr_label_logprob_fwd")
        ctx.save_for_backward(lg, idx, dist)
        ctx.ls, ctx.sum_scale = float(ls_weight), float(sum_scale)
        ctx.set_materialize_grads(False)
        if amax is not None:
            ctx.mark_non_differentiable(amax)
        return out, total, amax

    @staticmethod
    def backward(ctx, g, gt, _gamax):
        lg, idx, dist = ctx.saved_tensors
        V = lg.shape[-1]
        rows = lg.numel() // V
        if g is None and gt is None:
            return (None,) * 7
        if g is None:
            gc, stride, scale = gt.contiguous(), 0, ctx.sum_scale       # d(total) alone: one scalar for every row
        else:
            gc, stride, scale = (g if gt is None else g + gt * ctx.sum_scale).contiguous(), 1, 1.0
        dz = torch.empty_like(lg)
        hb.check(hb.load().asr_label_logprob_bwd(rows, V, hb.ptr(lg), V, ctypes.c_void_p(idx.data_ptr()), hb.ptr(dist),
                                                 ctx.ls, hb.ptr(gc), stride, scale, hb.ptr(dz), V, hb.stream()),
                 "asr_label_logprob_bwd")
        return (dz,) + (None,) * 6


def label_logprob(logits, index, labeldist=None, ls_weight=0.0, with_sum=False, sum_scale=1.0, with_argmax=False):
    """-> out, + total (with_sum), + argmax (with_argmax)."""
    out, total, amax = _LabelLogProb.apply(logits, index, labeldist, ls_weight, with_sum, sum_scale, with_argmax)
    res = (out,) + ((total,) if with_sum else ()) + ((amax,) if with_argmax else ())
    return res if len(res) > 1 else out


def decoder_sequence(P, Q, emb_w, w_ih, w_hh, b_ih, b_hh, wdec, convw, watt, gvec, bo, w_out, b_out, w0, opts):
    opts = dict(opts)
    opts["pooled"] = torch.is_grad_enabled() and (P.requires_grad or w_hh.requires_grad)
    return _DecoderSeq.apply(P, Q, emb_w, w_ih, w_hh, b_ih, b_hh, wdec, convw, watt, gvec, bo, w_out, b_out, w0, opts)


def attention_step(enc_pad, P, Q, wdec, convw, watt, gvec, bo, dec_z, att_prev, scaling):
    """One stand-alone location-aware attention step (AttLoc.forward, model.py:139-173) on the same kernels as the
    fused loop; forward only (the training path differentiates through decoder_sequence).
    Returns (mlp_o(context) [B,O], w [B,T'])."""
    dev = P.device
    B, Tp, A = P.shape
    O, D, C = Q.shape[2], wdec.shape[1], convw.shape[0]
    K = (convw.shape[-1] - 1) // 2
    E = 16                                                   # dummy embedding width (unused columns of X)
    ws = _dec_workspace(B, Tp, A, D, O, E, C, K, 1, False, dev, False)
    with torch.no_grad():
        ws["P"].copy_(P); ws["Q"].copy_(Q); ws["w0"].copy_(att_prev)
        ws["convw"].copy_(convw.reshape(C, 2 * K + 1)); ws["gvec"].copy_(gvec.reshape(A)); ws["wattT"].copy_(watt.t())
        ws["wcat"].zero_(); ws["bcat"].zero_(); ws["X"].zero_()
        ws["X"][1, :, :D] = dec_z
        d = dict(B=B, Tp=Tp, A=A, D=D, O=O, E=E, C=C, K=K, L=1, KX=D + O + E, scaling=float(scaling),
                 bo=bo.contiguous(), wdec=wdec.contiguous(), watt=watt.contiguous())
        d.update({k: ws[k] for k in ("P", "Q", "wcat", "bcat", "convw", "gvec", "wattT", "w0", "xmask", "X", "Xd",
                                     "gates", "cstate", "Dproj", "fconv", "S", "energy", "ws")})
        fs = _dec_fwd_struct(d, 0, B)
        hb.check(hb.load().asr_att_step_fwd(ctypes.byref(fs), 0, hb.stream()), "asr_att_step_fwd")
        return ws["X"][1, :, D:D + O].clone(), ws["ws"][0].clone()
