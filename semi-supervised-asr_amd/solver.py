"""Training driver with the reference Solver's surface (solver.py:13-565): same constructor, method
names, config keys, loss formulas and op order (zero_grad -> backward -> [all-reduce] -> clip -> step).
Differences, all MI355X-first: the optimiser is the fused flat-buffer FlatAdam (one HIP kernel for
clip+Adam(amsgrad)), gradients of a data-parallel run are exchanged with ONE RCCL all-reduce, and
under torch.distributed every rank draws the same global batch (identically seeded samplers), pads and uploads
only its strided shard - at the global extents (parallel.py), which reproduces the single-process numbers
exactly.  Batches reach HBM through feed.DeviceFeed: collated one step ahead into pinned memory and uploaded on
a side stream, so that the loops below never wait for a copy (the reference: to_gpu(data), utils.py:154-158).
"""
import math
import os
import pickle

import numpy as np
import torch

import hip_backend as hb
import ops
import parallel
from dataloader import get_data_loader, _raw_items, _raw_texts
from dataset import PickleDataset
from feed import DeviceFeed
from model import E2E, LM
from parallel import FlatAdam
from utils import Logger, adjust_learning_rate, calculate_cer, cc, infinite_iter, remove_pad_eos, to_sents


class StepScalar(object):
    """A scalar of a train step (loss, ...) whose host copy may still be on its way: the step's kernels, its optimiser
    update included, are enqueued without waiting for it (Solver._step).  float() / format() / arithmetic resolve it -
    which waits for that step and deals with an abort latch it may carry.  Under data parallelism resolving a record may
    turn into a sequence of collectives (the coordinated repeat, parallel.DpPipeline._recover), so every rank has to
    resolve at the same points of its program: the Solver's loops do (they run the same code on all ranks); code of its own
    that reads a scalar on ONE rank only (a rank-0 log, a rank-0 save) calls Solver.flush() on EVERY rank first."""
    __slots__ = ("_rec", "_i")

    def __init__(self, rec, i):
        self._rec, self._i = rec, i

    def __float__(self):
        if self._rec["values"] is None:
            self._rec["resolve"](self._rec)
        return float(self._rec["values"][self._i])

    item = __float__

    def __format__(self, spec):
        return format(float(self), spec)

    def __repr__(self):
        return repr(float(self))

    def __reduce__(self):                               # pickles (torch.save of a log) as the plain float
        return (float, (float(self),))


def _float_op(name):
    def op(self, *args):
        return getattr(float(self), name)(*[float(a) if isinstance(a, StepScalar) else a for a in args])
    return op


for _name in ("__add__", "__radd__", "__sub__", "__rsub__", "__mul__", "__rmul__", "__truediv__", "__rtruediv__", "__neg__",
              "__abs__", "__lt__", "__le__", "__gt__", "__ge__", "__eq__", "__ne__", "__pow__", "__bool__"):
    setattr(StepScalar, _name, _float_op(_name))
StepScalar.__hash__ = lambda self: hash(float(self))


class Solver(object):
    def __init__(self, config, load_model=False):
        self.config = config
        self._paths_reported = False
        self._pending = []                 # train steps whose host read (scalars + abort latch) is outstanding, oldest first
        self._pinned = None                # their landing slots in pinned host memory
        self._dp_pipe = None               # the same for data-parallel steps (parallel.DpPipeline)
        self.rank, self.world, _ = parallel.init_distributed()
        # The reference never seeds numpy (teacher-forcing draws model.py:328, input noise solver.py:370-373).  Data-parallel
        # ranks must draw identical streams, so `numpy_seed` (not a reference key) defaults to 0 there;
        # a single process stays unseeded like the reference unless the key is given.
        seed = config.get("numpy_seed", 0 if self.world > 1 else None)
        if seed is not None:
            np.random.seed(int(seed))
        if self.rank == 0:
            print(self.config)
        self.logger = Logger(config["logdir"])
        self.load_vocab()
        self.get_data_loaders()
        self.labeldist = self.get_label_dist(self.train_lab_dataset)
        self.unlab_labeldist = self.get_label_dist(self.train_unlab_y_dataset)
        self.proportion = self.calculate_length_proportion()
        self.build_model(load_model=load_model)
        self.settle_host_memory()          # the corpora stay for good: keep the garbage collector from walking them mid-epoch

    # ------------------------------------------------------------------ checkpoints (.ckpt/.opt/.judge.*)
    def save_model(self, model_path):
        self.flush()
        if self.rank == 0:
            torch.save(self.model.state_dict(), f"{model_path}.ckpt")
            torch.save(self.gen_opt.state_dict(), f"{model_path}.opt")

    def save_judge(self, model_path):
        self.flush()
        if self.rank == 0:
            torch.save(self.judge.state_dict(), f"{model_path}.judge.ckpt")
            torch.save(self.dis_opt.state_dict(), f"{model_path}.judge.opt")

    def load_model(self, model_path, load_optimizer):
        print(f"Load model from {model_path}.ckpt")
        self.model.load_state_dict(torch.load(f"{model_path}.ckpt", map_location="cpu"))
        if load_optimizer:
            print(f"Load optmizer from {model_path}.opt")
            self.gen_opt.load_state_dict(torch.load(f"{model_path}.opt", map_location="cpu"))

    def load_judge(self, model_path, load_optimizer):
        self.judge.load_state_dict(torch.load(f"{model_path}.judge.ckpt", map_location="cpu"))
        if load_optimizer:
            self.dis_opt.load_state_dict(torch.load(f"{model_path}.judge.opt", map_location="cpu"))

    # ------------------------------------------------------------------ data
    def load_vocab(self):
        with open(self.config["vocab_path"], "rb") as f:
            self.vocab = pickle.load(f)
        with open(self.config["non_lang_syms_path"], "rb") as f:
            self.non_lang_syms = pickle.load(f)

    def get_label_dist(self, dataset):
        counts = np.zeros(len(self.vocab))
        for _, tokens in dataset:
            np.add.at(counts, np.asarray(tokens, dtype=np.int64), 1.0)
        counts[self.vocab["<EOS>"]] += len(dataset)
        counts[self.vocab["<PAD>"]] = 0
        counts[self.vocab["<BOS>"]] = 0
        return counts / counts.sum()

    def calculate_length_proportion(self):
        frames = sum(x.shape[0] for x, _ in self.train_lab_dataset)
        chars = sum(len(y) for _, y in self.train_lab_dataset)
        return chars / frames

    def _dataset(self, name, config, sort=True):
        return PickleDataset(os.path.join(self.config["dataset_root_dir"], f"{name}.pkl"), config=config, sort=sort)

    def _loader(self, dataset, batch_size, shuffle, drop_last, **kw):
        # every rank must draw the SAME global batches: seed the sampler identically
        gen = torch.Generator()
        gen.manual_seed(int(self.config.get("data_seed", 0)))
        return get_data_loader(dataset, batch_size=batch_size, shuffle=shuffle, drop_last=drop_last, generator=gen,
                               **kw)

    def get_data_loaders(self):
        cfg = self.config
        bs, shuffle = cfg["batch_size"] * self.world, cfg["shuffle"]     # batch_size is per GPU
        bucket = bool(cfg.get("bucket_batches", False))     # not a reference key: length-bucketed speech batches
        self.train_lab_dataset = self._dataset(cfg["labeled_set"], cfg)
        self.train_lab_loader = self._loader(self.train_lab_dataset, bs, shuffle, False, bucket=bucket)
        self.train_unlab_x_dataset = self._dataset(cfg["unlabeled_speech_set"], cfg)
        self.train_unlab_x_loader = self._loader(self.train_unlab_x_dataset, bs, shuffle, False, speech_only=True,
                                                 bucket=bucket)
        self.train_unlab_y_dataset = self._dataset(cfg["unlabeled_text_set"], cfg)
        self.train_unlab_y_loader = self._loader(self.train_unlab_y_dataset, bs, shuffle, True, text_only=True)
        self.dev_dataset = self._dataset(cfg["dev_set"], None)
        self.dev_loader = self._loader(self.dev_dataset, cfg["batch_size"] // 2, False, False)

    def _feed(self, loader, kind="labeled", sharded=True, noise_std=0.0, endless=False):
        """The device-side view of a loader: the same batches in the same order (the loader's own batch sampler), collated
        one step ahead into pinned memory and uploaded on a side stream (feed.DeviceFeed) - this rank's strided rows only
        when `sharded` and the run is data parallel.  Config keys (not reference keys): `prefetch_batches` (default 2),
        `prefetch_thread` (default true: collate in a background thread)."""
        from torch.utils.data import DataLoader
        # (the loader's generator too: iterating a DataLoader draws a base seed from it before the sampler draws its
        # permutation - the batches are then the ones `for data in loader` would have produced)
        raw = DataLoader(loader.dataset, batch_sampler=loader.batch_sampler, num_workers=0, generator=loader.generator,
                         collate_fn=_raw_texts if kind == "text" else _raw_items)
        rank, world = (self.rank, self.world) if sharded else (0, 1)
        return DeviceFeed(infinite_iter(raw) if endless else raw, "cuda" if torch.cuda.is_available() else "cpu", kind=kind,
                          rank=rank, world=world, depth=int(self.config.get("prefetch_batches", 2)), noise_std=noise_std,
                          thread=bool(self.config.get("prefetch_thread", True)))

    def get_infinite_iter(self):
        self.lab_iter = iter(self._feed(self.train_lab_loader, endless=True))
        self.unlab_x_iter = iter(self._feed(self.train_unlab_x_loader, kind="speech", endless=True))
        self.unlab_y_iter = iter(self._feed(self.train_unlab_y_loader, kind="text", endless=True))

    # ------------------------------------------------------------------ model + optimisers
    def build_model(self, load_model=False):
        cfg = self.config
        self.model = cc(E2E(
            input_dim=cfg["input_dim"], enc_hidden_dim=cfg["enc_hidden_dim"], enc_n_layers=cfg["enc_n_layers"],
            subsample=cfg["subsample"], dropout_rate=cfg["dropout_rate"], dec_hidden_dim=cfg["dec_hidden_dim"],
            att_dim=cfg["att_dim"], conv_channels=cfg["conv_channels"], conv_kernel_size=cfg["conv_kernel_size"],
            att_odim=cfg["att_odim"], output_dim=len(self.vocab), embedding_dim=cfg["embedding_dim"],
            ls_weight=cfg["ls_weight"], labeldist=self.labeldist, pad=self.vocab["<PAD>"], bos=self.vocab["<BOS>"],
            eos=self.vocab["<EOS>"]))
        # `dp_overlap` (not a reference key): issue the gradient all-reduce in buckets from inside the backward pass
        # (parallel.FlatBuffers.enable_overlap).  Off by default HERE: an RCCL kernel that is resident while a persistent
        # kernel is being placed could - if the two do not fit a CU together and a rank is late - hold that kernel's
        # workgroups back until its bounded spins expire (every rank then repeats the step off the persistent kernels).
        # bench.py measures it as a labelled sub-object; turn it on here once a multi-GPU run has shown the latch stays clear.
        overlap = bool(cfg.get("dp_overlap", False))
        # `persist_retry_steps` (not a reference key): train steps on the per-step kernels after an abort of the persistent
        # ones before they are tried again (hip_backend.PERSIST_RETRY_STEPS: 200, doubling per abort; 0: never)
        if "persist_retry_steps" in cfg:
            hb.PERSIST_RETRY_STEPS = int(cfg["persist_retry_steps"])
        self.gen_opt = FlatAdam(self.model, lr=cfg["learning_rate"], weight_decay=cfg["weight_decay"], amsgrad=True,
                                max_grad_norm=cfg["max_grad_norm"], overlap=overlap)
        if load_model:
            self.load_model(cfg["load_model_path"], cfg["load_optimizer"])
        self.judge = cc(LM(
            output_dim=len(self.vocab), embedding_dim=cfg["dis_embedding_dim"], hidden_dim=cfg["dis_hidden_dim"],
            dropout_rate=cfg["dis_dropout_rate"], n_layers=cfg["dis_layers"], bos=self.vocab["<BOS>"],
            eos=self.vocab["<EOS>"], pad=self.vocab["<PAD>"], ls_weight=cfg["ls_weight"],
            labeldist=self.unlab_labeldist))
        self.dis_opt = FlatAdam(self.judge, lr=cfg["d_learning_rate"], max_grad_norm=cfg["max_grad_norm"], overlap=overlap)
        if self.rank == 0:
            print(self.model)
            print(self.judge)

    # ------------------------------------------------------------------ text scoring
    def ind2sent(self, all_prediction, all_ys):
        hyp_ids = remove_pad_eos(all_prediction, eos=self.vocab["<EOS>"])
        hyps = to_sents(hyp_ids, self.vocab, self.non_lang_syms)
        refs = to_sents(all_ys, self.vocab, self.non_lang_syms)
        return calculate_cer(hyps, refs), hyps, refs

    def _abort_seen(self, dev):
        """True if a persistent kernel aborted on ANY rank since the latch was last cleared (synchronises).  Under data
        parallelism the decision is collective (all-reduce MAX of the latch), so that all ranks leave the persistent
        kernels together or none does - ranks on different kernel paths would still compute the same numbers, but a rank
        that alone repeats a batch must not be the one the others wait for in the next collective."""
        if dev.type != "cuda":
            return False
        flag = hb.persist_abort_flag(dev)[:1].float().clone()
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        return flag.item() != 0.0

    def _greedy(self, xs, ilens):
        """Greedy hypothesis ids for one batch.  No loss is read on this path, so the abort word of the persistent
        kernels is checked explicitly after the decode (the copy to the host has synchronised anyway): an aborted
        launch poisons its outputs, which would otherwise surface as a silently wrong CER.  On an abort this process
        switches to the per-step kernels and decodes the batch again, as _backward_guarded does for a train step."""
        def run():
            with torch.no_grad():
                _, _, prediction, _ = self.model(xs, ilens, ys=None, max_dec_timesteps=self.config["max_dec_timesteps"])
            return prediction.cpu().numpy().tolist()
        out = run()
        if self._abort_seen(xs.device):
            print("persistent kernels aborted during greedy decoding (this rank's code %d): repeating the batch on the "
                  "per-step kernels" % hb.persist_abort_code(xs.device))
            hb.disable_persistent(xs.device)
            out = run()
        return out

    def validation(self):
        """Teacher-forced dev loss + greedy CER (solver.py:212-242); greedy pass runs without autograd."""
        self.flush()
        self.model.eval()
        preds, refs, total = [], [], 0.0
        for batch in self._feed(self.dev_loader, sharded=False):
            xs, ilens, ys = batch
            with torch.no_grad():
                _, log_probs, _, _ = self.model(xs, ilens, ys=ys)
                value = self.model.mask_and_cal_loss(log_probs, ys).item()
                if self._abort_seen(xs.device):             # the latch, not the loss: an abort need not reach the loss
                    hb.disable_persistent(xs.device)        # see _greedy
                    _, log_probs, _, _ = self.model(xs, ilens, ys=ys)
                    value = self.model.mask_and_cal_loss(log_probs, ys).item()
                total += value
            preds += self._greedy(xs, ilens)
            refs += batch.ys_host
        self.model.train()
        cer, hyp_sents, ref_sents = self.ind2sent(preds, refs)
        return total / len(self.dev_loader), cer, hyp_sents, ref_sents

    def lm_validation(self):
        self.flush()
        self.judge.eval()
        total = 0.0
        for batch in self._feed(self.dev_loader, kind="text", sharded=False):     # transcripts, longest first
            ys = list(batch)
            with torch.no_grad():
                log_probs, _, _ = self.judge(ys)
                value = -self.judge.mask_and_cal_sum(log_probs, ys).item()
                if ys and self._abort_seen(ys[0].device):   # the H = 640 judge runs on the persistent kernels too: a latch
                    hb.disable_persistent(ys[0].device)     # left here would fail the next (clean) training step
                    log_probs, _, _ = self.judge(ys)
                    value = -self.judge.mask_and_cal_sum(log_probs, ys).item()
                total += value
        samples = self.judge.decode(n_samples=5, sample=True,
                                    max_dec_timesteps=int(self.config.get("lm_sample_steps", 100)))
        sents = to_sents(remove_pad_eos(samples.cpu().numpy(), eos=self.vocab["<EOS>"]), self.vocab,
                         self.non_lang_syms)
        self.judge.train()
        return total / len(self.dev_loader), sents

    def test(self, state_dict=None):
        if state_dict:
            self.model.load_state_dict(state_dict)
        else:
            self.load_model(self.config["load_model_path"], self.config["load_optimizer"])
        test_set = self.config["test_set"]
        loader = get_data_loader(self._dataset(test_set, None, sort=False), batch_size=1, shuffle=False,
                                 drop_last=False)
        self.model.eval()
        preds, refs = [], []
        for batch in self._feed(loader, sharded=False):
            xs, ilens, _ = batch
            preds += self._greedy(xs, ilens)
            refs += batch.ys_host
        self.model.train()
        cer, hyp_sents, _ = self.ind2sent(preds, refs)
        with open(f"{test_set}.txt", "w") as f:
            f.writelines(f"{p}\n" for p in hyp_sents)
        print(f"{test_set}: {len(hyp_sents)} utterances, CER={cer:.4f}")
        return cer

    # ------------------------------------------------------------------ judge (LM) pre-training
    # Number of train steps whose host read may be outstanding when the next one is enqueued.  1 (default): the host looks
    # at step i's loss and abort latch while step i + 1 runs - no GPU idle time between steps.  0: the reference's order,
    # loss.item() inside every step (solver.py:379).  Config key `pipeline_steps` (not a reference key).
    PIPELINE_STEPS = 1
    PINNED_ROWS = 8                    # landing rows for the host records of outstanding steps: pipeline_steps <= PINNED_ROWS - 2

    def _step(self, make_local, opt, n_scalars):
        """Run one optimiser step on make_local() -> (local loss, [scalar tensors]); returns the scalars, summed over the
        ranks of a data-parallel run (= the single-process values), as StepScalars (plain floats from the synchronous
        data-parallel step, `pipeline_steps: 0`).

        One process: nothing in the step waits for the host.  zero_grad -> backward -> gradients into the flat buffer ->
        [loss scalars + abort latch -> pinned host memory, asynchronously] -> clip + Adam, whose kernel checks the abort latch
        ON THE DEVICE and does nothing when it is set (asr_adam_clip_f32: skip_if_nonzero).  The host reads a step's record
        while the NEXT step runs (the reference's per-step loss.item(), one step late).  The persistent XCD-local kernels
        poison their outputs with NaN and set the sticky latch when they abort (a workgroup placement other than one per CU,
        a bounded spin expiring: a shared or partitioned GPU); the latch stays set, so the update of the aborted step AND of
        every step enqueued behind it was skipped on the device - _recover switches this process to the per-step kernels
        and runs those steps again from the numpy stream (teacher-forcing draws) the first of them started with.  A
        non-finite loss without the latch is the model's own and is applied, as in the reference.

        Data parallel: the same, with the latch SUMMED OVER THE RANKS in the step's one all-reduce (last aux slot of the flat
        buffer) as the device-side predicate - every rank skips or none does - and the coordinated repeat one step late
        (parallel.DpPipeline).  Every rank resolves its StepScalars at the same points of the program (the loops
        below run the same code on all ranks): a recovery is a sequence of collectives."""
        dev0 = opt.buf.flat_g.device
        if hb.persistent_step_tick() and self.rank == 0:     # the end of a probation after an abort (hb.PERSIST_RETRY_STEPS)
            print("persistent kernels: trying them again after %d abort(s)" % hb.persistent_probation()[0])
        with ops.step_arena(dev0):       # every zero-initialised accumulator of the step comes out of one buffer, one fill
            return self._step_inner(make_local, opt, n_scalars)

    def _step_inner(self, make_local, opt, n_scalars):
        depth = min(int(self.config.get("pipeline_steps", self.PIPELINE_STEPS)), self.PINNED_ROWS - 2)   # one landing row each
        if self.world > 1 or parallel.FORCE_DP:
            pipe = self._dp_pipeline(depth, opt.buf.flat_g.device)
            rec = pipe.step(make_local, opt, n_scalars)
            if depth <= 0:
                pipe.resolve(rec)                        # `pipeline_steps: 0`: the same path, read at once
            self._report_paths()
            return [StepScalar(rec, i) for i in range(n_scalars)]
        rec = dict(resolve=self._resolve_through, opt=opt, make_local=make_local, n=n_scalars, rng=np.random.get_state(),
                   values=None, event=None, slot=None)
        loss, scalars = make_local()
        opt.zero_grad()
        parallel.backward(loss)                          # (the seed of the backward pass is a cached tensor)
        opt.reduce()                                     # gradients -> flat buffer (no collective in one process)
        if not loss.is_cuda:                             # CPU tensors (tests of the host logic): nothing to pipeline
            opt.apply()
            rec["values"] = [float(v) for v in scalars[:n_scalars]]
            return [StepScalar(rec, i) for i in range(n_scalars)]
        dev = loss.device
        latch = hb.persist_abort_flag(dev)
        rec["slot"] = self._free_slot()
        stage = torch.stack([v.detach().reshape(()).float() for v in scalars[:n_scalars]] + [latch[0].float()])
        self._pinned[rec["slot"], :n_scalars + 1].copy_(stage, non_blocking=True)
        rec["event"] = torch.cuda.Event()
        rec["event"].record()
        opt.apply(skip_if=latch)                         # clip -> Adam; a no-op on the device if the latch is set
        self._pending.append(rec)
        while len(self._pending) > depth:
            self._resolve_through(self._pending[0])
        self._report_paths()
        return [StepScalar(rec, i) for i in range(n_scalars)]

    @staticmethod
    def settle_host_memory():
        """The corpora, vocabularies and loaders built so far live as long as the process: collect what is garbage now and move
        the rest out of the garbage collector's sight (gc.freeze).  A full collection that has to walk a corpus of a few
        thousand utterances (dicts of arrays and token lists) takes tens of milliseconds - and when it strikes in the middle
        of an epoch the host, one step ahead of the GPU, falls behind it (measured: a 60-batch epoch at cfg-2 ran 12.6 ms
        per step instead of 11.7, one 50 ms pause)."""
        import gc
        gc.collect()
        gc.freeze()

    def _free_slot(self):
        """A row of the pinned landing buffer that no outstanding step uses."""
        if self._pinned is None:
            self._pinned = torch.zeros(self.PINNED_ROWS, 8, dtype=torch.float32).pin_memory()
        busy = set(r["slot"] for r in self._pending)
        if self._dp_pipe is not None:
            row = self._pinned.stride(0) * self._pinned.element_size()
            busy |= set((r["host"].data_ptr() - self._pinned.data_ptr()) // row for r in self._dp_pipe.pending)
        return next(i for i in range(self._pinned.shape[0]) if i not in busy)

    def _dp_pipeline(self, depth, dev):
        """The data-parallel step (parallel.DpPipeline, the only one): the abort latch rides in the last aux slot of the step's
        ONE all-reduce, its sum predicates the update on the device - every rank skips or none does -, the reduced scalars
        land in pinned host memory behind an event; when a record shows the latch set EVERY rank leaves the persistent
        kernels (a rank that alone repeats a batch would be one collective the others do not take part in) and repeats what
        was skipped, from the numpy streams those steps started with."""
        if self._dp_pipe is None and dev.type != "cuda":           # (CPU tensors: tests of the host logic on gloo)
            self._dp_pipe = parallel.DpPipeline(max(1, depth), lambda: 0.0, lambda n: None)
        if self._dp_pipe is None:
            def stage(aux):
                host = self._pinned[self._free_slot(), :aux.numel()]
                host.copy_(aux, non_blocking=True)
                event = torch.cuda.Event()
                event.record()
                return host, event.synchronize

            def latch():
                return hb.persist_abort_flag(torch.device("cuda", torch.cuda.current_device()))[0].float()

            def leave_persistent(n_ranks):
                dev = torch.device("cuda", torch.cuda.current_device())
                torch.cuda.synchronize()
                print("rank %d: persistent kernels aborted on %d rank(s) (this rank: %s, code %d): every rank repeats the "
                      "skipped step(s) on the per-step kernels" % (self.rank, n_ranks, hb.persist_aborted(dev),
                                                                   hb.persist_abort_code(dev)))
                hb.disable_persistent(dev)              # also clears this rank's latch
            self._free_slot()                           # allocates the pinned buffer
            self._dp_pipe = parallel.DpPipeline(depth, latch, leave_persistent, stage)
        self._dp_pipe.depth = max(1, int(depth))
        return self._dp_pipe

    def _report_paths(self):
        if not self._paths_reported:
            # which kernels the sequence operators of the first step ran on (hb.LAUNCHES: *_persist = the persistent
            # XCD-local kernels, *_step = the per-step kernels, 3x slower: unsupported width, shared or partitioned GPU)
            self._paths_reported = True
            if self.rank == 0:
                print("sequence-operator paths of the first step: %s (persistent kernels %s)"
                      % (dict(sorted(hb.LAUNCHES.items())), "on" if hb.USE_PERSIST else "off"))

    def _resolve_through(self, rec):
        """Read the host records of every outstanding step up to and including `rec` (oldest first)."""
        while self._pending and rec["values"] is None:
            first = self._pending[0]
            first["event"].synchronize()
            vals = self._pinned[first["slot"], :first["n"] + 1].tolist()
            if vals[-1] != 0.0:
                self._recover()
            else:
                first["values"] = vals[:first["n"]]
                first["make_local"] = first["opt"] = None   # a StepScalar kept in a log must not keep the batch alive
                self._pending.pop(0)

    def _lagged(self, log):
        """-> (push, done): push(item) reports the PREVIOUS item through log(item) - its step's host record has landed by
        then, so printing its loss does not make the host wait for the step that was just enqueued - and done() reports
        the last one (the reference prints every step's loss inside the step, solver.py:379-381)."""
        held = []

        def push(item):
            if held:
                log(held.pop())
            held.append(item)

        def done():
            if held:
                log(held.pop())
        return push, done

    def flush(self):
        """Wait for the host records of all outstanding train steps (before validation, checkpoints, the end of a loop)."""
        while self._pending:
            self._resolve_through(self._pending[-1])
        if self._dp_pipe is not None:
            self._dp_pipe.flush()

    def _recover(self):
        """The oldest outstanding step found the abort latch set: neither its update nor that of any step enqueued behind it
        was applied (the latch is sticky and the Adam kernel checks it on the device).  Leave the persistent kernels (until
        the probation of hip_backend.persistent_step_tick ends), take the optimisers' step counts back and run those steps
        again, in order, each from the numpy stream it started with and in an arena scope of its own (this may run at the end
        of the step that found the abort, inside ITS scope: that step is fully enqueued by then and hands back scalars only,
        so its slices may be zeroed - ops._Arena, LIFETIME)."""
        torch.cuda.synchronize()
        redo, self._pending = self._pending, []
        dev = redo[0]["opt"].buf.flat_g.device
        print("persistent kernels aborted (code %d): continuing on the per-step kernels, repeating %d step(s)"
              % (hb.persist_abort_code(dev), len(redo)))
        hb.disable_persistent(dev)                       # also clears the latch
        for rec in redo:
            rec["opt"].unapply()
        resume = np.random.get_state()                   # where the stream stands now: behind every draw made so far
        for rec in redo:
            # each step again from the stream state IT started with: draws made between the steps (none in the loops of
            # this file - the input noise has a stream of its own, feed.DeviceFeed - but a caller may make some) are not
            # part of a step and must not shift the teacher-forcing draws of the repeats
            np.random.set_state(rec["rng"])
            with ops.step_arena(dev):
                loss, scalars = rec["make_local"]()
                rec["opt"].zero_grad()
                loss.backward()
                both = torch.stack([v.detach().reshape(()).float() for v in scalars[:rec["n"]]] +
                                   [hb.persist_abort_flag(dev)[0].float()]).tolist()
                if both[-1] != 0.0:
                    raise RuntimeError("the abort latch is set (code %d) after a step on the per-step kernels"
                                       % hb.persist_abort_code(dev))
                rec["opt"].step()
            rec["values"] = both[:rec["n"]]
            rec["make_local"] = rec["opt"] = None
        np.random.set_state(resume)                      # a step draws the same number of values whatever its outputs were

    def judge_train_one_iteration(self, unlab_ys):
        """solver.py:288-301.  `unlab_ys` is the global text batch - every rank of a data-parallel run takes its strided
        shard and normalises by the global sum of (len + 5) (parallel.judge_local_loss; the identity for one process) - or
        this rank's parallel.LocalShard of it (the loops below: only the shard was uploaded)."""
        def make_local():
            loss, avg_prob = parallel.judge_local_loss(
                lambda ys: self.judge(ys=ys, discrete_input=True),
                lambda v, ys: self.judge.mask_and_cal_sum(v, ys=ys, mask=None), unlab_ys, self.rank, self.world)
            return loss, [loss, avg_prob]
        value, avg_prob = self._step(make_local, self.dis_opt, 2)
        return {"loss": value, "avg_prob": avg_prob}

    def judge_pretrain(self):
        cfg = self.config
        steps_per_epoch = len(self.train_unlab_y_loader)
        best = 100
        base_lr = cfg["d_learning_rate"]
        path = os.path.join(cfg["model_dir"], cfg["model_name"])
        print("--------Judge pretraining--------")
        for epoch in range(cfg["judge_epochs"]):
            # MultiStepLR(milestones=[dis_change_learning_rate_epoch], gamma) stepped at epoch start (solver.py:308-315)
            adjust_learning_rate(self.dis_opt, base_lr * (cfg["lr_gamma"] if epoch + 1 >= cfg[
                "dis_change_learning_rate_epoch"] else 1.0))
            total = [0.0]

            def log(item):
                it, meta = item
                total[0] += float(meta["loss"])
                print(f"epoch: {epoch}, [{it + 1}/{steps_per_epoch}], loss: {meta['loss']:.3f}, "
                      f"prob: {meta['avg_prob']:.3f}", end="\r")
                for key, val in meta.items():
                    self.logger.scalar_summary(f"{cfg['tag']}/judge_pretrain/{key}", float(val),
                                               epoch * steps_per_epoch + it + 1)
            push, done = self._lagged(log)
            for it, batch in enumerate(self._feed(self.train_unlab_y_loader, kind="text")):
                push((it, self.judge_train_one_iteration(batch.xs if self.world > 1 else batch.ys)))
            done()
            running = total[0]
            val_loss, samples = self.lm_validation()
            print(f"epoch: {epoch}, train_loss={running / steps_per_epoch:.3f}, valid_loss={val_loss:.3f}")
            for i, s in enumerate(samples):
                print(f"hyp-{i + 1}: {s}")
            self.logger.scalar_summary(f"{cfg['tag']}/judge_pretrain/val_loss", val_loss, epoch)
            self.logger.scalar_summary(f"{cfg['tag']}/judge_pretrain/avg_train_loss", running / steps_per_epoch, epoch)
            if val_loss < best:
                best = val_loss
                self.save_judge(path)
                print(f"save #{epoch} LM, val_loss={val_loss:.3f}")
            self.save_judge(f"{path}-{epoch:03d}")

    # ------------------------------------------------------------------ supervised training
    def _sharded_forward(self, xs, ilens, ys, tf_rate):
        """Model forward on this rank's strided shard of the (global) batch, padded to the global extents;
        returns the local loss whose all-reduced gradient equals the single-process one; None for a rank
        whose shard is empty (fewer utterances than ranks in the last batch of an epoch).  One process: the whole
        batch and -mean(log_probs) (solver.py:375-378)."""
        cfg = self.config
        return parallel.sup_local_loss(self.model, xs, ilens, ys, tf_rate, self.rank, self.world,
                                       cfg["enc_n_layers"], cfg["subsample"])

    def sup_train_one_iteration(self, xs, ilens, ys, tf_rate):
        """The body of the supervised loop (solver.py:375-385) for one batch already on the device - the global batch, or this
        rank's parallel.LocalShard of it as `xs` (what the epoch loop's feed uploads under data parallelism): forward on this
        rank's shard, loss = -mean(log_probs), zero_grad, backward, (all-reduce,) loss + abort latch read on the host - the
        reference's loss.item() - then clip + Adam.  bench.py times exactly this method."""
        def make_local():
            loss = self._sharded_forward(xs, ilens, ys, tf_rate)
            return loss, [loss]
        value, = self._step(make_local, self.gen_opt, 1)         # (all-reduce ->) clip -> Adam; local losses sum to the mean
        return value

    def sup_train_one_epoch(self, epoch, tf_rate):
        cfg = self.config
        steps_per_epoch = len(self.train_lab_loader)
        running = [0.0]

        def log(item):
            it, value = item
            running[0] += float(value)
            if self.rank == 0:
                print(f"epoch: {epoch}, [{it + 1}/{steps_per_epoch}], loss: {value:.3f}", end="\r")
            self.logger.scalar_summary(tag=f"{cfg['tag']}/train_loss", value=float(value),
                                       step=epoch * steps_per_epoch + it + 1)
        push, done = self._lagged(log)
        # input noise (solver.py:370-373) is added by the feed, on the host, before the upload
        noise = float(cfg["gaussian_std"]) if cfg["add_gaussian"] and epoch >= cfg["gaussian_epoch"] else 0.0
        for it, (xs, ilens, ys) in enumerate(self._feed(self.train_lab_loader, noise_std=noise)):
            push((it, self.sup_train_one_iteration(xs, ilens, ys, tf_rate)))
        done()
        return running[0] / steps_per_epoch

    def sup_pretrain(self):
        cfg = self.config
        self.model.train()
        best_cer, best_model = 200, None
        hi, lo, span = cfg["init_tf_rate"], cfg["tf_rate_lowerbound"], cfg["tf_decay_epochs"]
        path = os.path.join(cfg["model_dir"], cfg["model_name"])
        print("------supervised pretraining-------")
        for epoch in range(cfg["epochs"]):
            tf_rate = hi - (hi - lo) * (epoch / span) if epoch <= span else lo
            train_loss = self.sup_train_one_epoch(epoch, tf_rate)
            val_loss, cer, hyps, refs = self.validation()
            print(f"Epoch: {epoch}, tf_rate={tf_rate:.3f}, train_loss={train_loss:.4f}, "
                  f"valid_loss={val_loss:.4f}, CER={cer:.4f}")
            tag = cfg["tag"]
            self.logger.scalar_summary(f"{tag}/supervised/cer", cer, epoch)
            self.logger.scalar_summary(f"{tag}/supervised/val_loss", val_loss, epoch)
            self.logger.scalar_summary(f"{tag}/supervised/avg_train_loss", train_loss, epoch)
            for i, (p, gt) in enumerate(list(zip(hyps, refs))[:5]):
                self.logger.text_summary(f"{tag}/supervised/prediction-{i}", p, epoch)
                self.logger.text_summary(f"{tag}/supervised/ground_truth-{i}", gt, epoch)
                print(f"hyp-{i + 1}: {p}\nref-{i + 1}: {gt}")
            if cer < best_cer:
                best_cer = cer
                self.save_model(path)
                best_model = self.model.state_dict()
                print(f"Save #{epoch} model, val_loss={val_loss:.3f}, CER={cer:.3f}")
            self.save_model(f"{path}-{epoch:03d}")
        return best_model, best_cer

    # ------------------------------------------------------------------ semi-supervised training
    def gen_train_one_iteration(self, lab_xs, lab_ilens, lab_ys, unlab_xs, unlab_ilens):
        """The LM-judge auxiliary loss (solver.py:460-495): greedy smooth-embedding decode of unlabeled
        speech WITH grad, judge probabilities of the hypothesis, unsup = -sum(p_LM * log p_model * mask)/sum(mask);
        loss = sup + unsup_weight * unsup; only the generator is stepped.
        lab_* / unlab_* are the GLOBAL batches (the loaders draw batch_size * world utterances) or this rank's
        parallel.LocalShards of them as lab_xs / unlab_xs: every rank works on its strided shards at the global padded extents and normalises the auxiliary loss by the global hypothesis-token
        count (parallel.ssl_local_loss: one 4-byte all-reduce next to the gradient all-reduce; nothing in one process)."""
        cfg = self.config

        def judge_probs(hyp):
            # The judge scores an integer hypothesis, so no gradient reaches the model through it, and gen_opt does not
            # hold its parameters (solver.py:484-488 builds that graph and never uses it): the forward alone gives the
            # same losses and model gradients.
            with torch.no_grad():
                return self.judge(ys=hyp, discrete_input=False)[1]

        def make_local():
            loss, (unsup, sup) = parallel.ssl_local_loss(
                self.model, judge_probs, (lab_xs, lab_ilens, lab_ys), (unlab_xs, unlab_ilens), self.rank, self.world,
                eos=self.vocab["<EOS>"], unsup_weight=cfg["unsup_weight"], proportion=self.proportion,
                smooth=cfg["smooth_embedding"], scaling=cfg["softmax_scaling"], n_layers=cfg["enc_n_layers"],
                subsample=cfg["subsample"])
            return loss, [unsup, sup, loss]
        unsup, sup, value = self._step(make_local, self.gen_opt, 3)
        return {"unsup_loss": unsup, "sup_loss": sup, "loss": value}

    def ssl_train_one_iteration(self, iteration):
        lab_xs, lab_ilens, lab_ys = next(self.lab_iter)
        unlab_xs, unlab_ilens = next(self.unlab_x_iter)
        return self.gen_train_one_iteration(lab_xs, lab_ilens, lab_ys, unlab_xs, unlab_ilens)

    def ssl_train(self):
        cfg = self.config
        print("--------SSL training--------")
        adjust_learning_rate(self.gen_opt, cfg["g_learning_rate"])
        best_cer, best_model = 2, None
        total = cfg["ssl_iterations"]
        path = os.path.join(cfg["model_dir"], cfg["model_name"])
        if not hasattr(self, "lab_iter"):
            self.get_infinite_iter()
        def log(item):
            it, meta = item
            print(f"[{it + 1}/{total}], sup_loss: {meta['sup_loss']:.3f}, unsup_loss: {meta['unsup_loss']:.3f}, "
                  f"loss: {meta['loss']:.3f}", end="\r")
            for key, val in meta.items():
                self.logger.scalar_summary(f"{cfg['tag']}/ssl_generator/{key}", float(val), it + 1)
        push, done = self._lagged(log)
        for step in range(total):
            push((step, self.ssl_train_one_iteration(iteration=step)))
            if (step + 1) % cfg["summary_steps"] == 0 or step + 1 == total:
                done()
                val_loss, cer, hyps, refs = self.validation()
                print(f"Iter: [{step + 1}/{total}], valid_loss={val_loss:.4f}, CER={cer:.4f}")
                self.logger.scalar_summary(f"{cfg['tag']}/ssl/cer", cer, step + 1)
                self.logger.scalar_summary(f"{cfg['tag']}/ssl/val_loss", val_loss, step + 1)
                for i, (p, gt) in enumerate(list(zip(hyps, refs))[:5]):
                    print(f"hyp-{i + 1}: {p}\nref-{i + 1}: {gt}")
                if cer < best_cer:
                    best_cer = cer
                    self.save_model(path)
                    self.save_judge(path)
                    best_model = self.model.state_dict()
                    print(f"Save #{step} model, val_loss={val_loss:.3f}, CER={cer:.3f}")
        return best_model, best_cer
