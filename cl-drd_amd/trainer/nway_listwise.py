"""Training step of the multistep-curriculum trainers (reference ``trainer/multistep-curriculum/nway_listwise_{1,2,3}.py``,
step loop ``nway_listwise_1.py:328-367``), MI355X-native and data-parallel over RCCL.

One process per GPU (``torch.distributed`` backend "nccl" == RCCL over xGMI).  Per step, on each rank:

    forward (query tower, passage tower) -> scoring -> loss value + dlogits (one kernel) -> hand-written backward
    -> per-layer gradient buckets all-reduced on a side stream while the backward of earlier layers still runs
    -> clip_grad_norm_(max_grad_norm) as one norm pass -> legacy-HF AdamW + bf16 shadows in one pass -> linear schedule

Differences from the reference, on purpose (SURVEY.md sections 7, 8a14):
  * mixed precision is hand-placed instead of autocast: every MFMA operand of a training pass is fp16 (the reference's own
    ``use_fp16`` mode; ``CLDRD_AMP=bf16`` runs the backward / tape on bf16 operands instead - there is NO all-bf16 mode: its forward FFN /
    out-projection / query tower stay fp16), fp32 accumulate, residual stream, gradient stream and master weights;
  * the loss scale is a power of two recomputed from dL/dCLS every step (``cldrd_loss_scale_adapt``), where ``GradScaler`` starts at
    65 536 and searches by overflowing: no early skipped steps here; what is kept is the safety net (a non-finite gradient norm skips the
    step, which - as in ``scaler.step`` - does not advance Adam's bias-correction step);
  * no per-step D2H sync: loss / train-MRR are read back only every ``logging_steps``;
  * the three stage scripts differ only in defaults, so there is one trainer with a ``--loss`` selector
    (default ``lambda_mrr`` as in the reference).
"""
from __future__ import annotations

import math
import os

import torch
import torch.distributed as dist

from .. import hip_ops as ops
from ..dataset.nway_dataset import attach_lengths
from ..encoder import _env_flag
from ..models.nway_dual_encoder import NwayDualEncoder, _lengths, score_mode
from ..retriever.retrieval_utils import cap_host_threads

LOSS_KINDS = ("lambda_mrr", "ranknet", "kl_div", "margin_mse")


def no_decay(name: str) -> bool:
    """reference nway_listwise_1.py:259-263 (substring match on 'bias' / 'LayerNorm.weight')."""
    return ("bias" in name) or ("LayerNorm.weight" in name)


def linear_schedule_factor(step: int, warmup_steps: int, total_steps: int) -> float:
    """``transformers.get_linear_schedule_with_warmup`` lambda (reference nway_listwise_1.py:265)."""
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    return max(0.0, float(total_steps - step) / float(max(1, total_steps - warmup_steps)))


def allreduce_buckets(flat_g: torch.Tensor, buckets, world: int, group=None):
    """SUM all-reduce of every (tower, layer, start, end) bucket of the flat gradient buffer (blocking form; the trainer
    issues the same calls per bucket from the backward hooks on a side stream).  The division by world_size is folded
    into dlogits, so the reduced buffer already holds the mean gradient."""
    if world <= 1:
        return
    for _, _, a, b in buckets:
        dist.all_reduce(flat_g[a:b], op=dist.ReduceOp.SUM, group=group)


def owns_example(line_idx: int, rank: int, nranks: int) -> bool:
    """Data sharding rule of the reference (dataset/nway_dataset.py:305): example line i belongs to rank i % nranks."""
    return line_idx % nranks == rank


def optimizer_param_groups(model):
    """Parameter names in the reference optimizer's index order: ``model.named_parameters()`` of the (DDP-wrapped)
    NwayDualEncoder - HF module order, a shared tower listed once under ``query_encoder`` - split into the decayed group, then
    the no-decay group (nway_listwise_1.py:259-263); torch numbers the parameters consecutively over the groups.
    Returns [[(full name, tower index or None for parameters this package does not hold (BERT pooler), HF name)]] per group."""
    from ..encoder import hf_parameter_order
    names = []
    for ti, (prefix, tower) in enumerate((("query_encoder", model.query_encoder), ("passage_encoder", model.passage_encoder))):
        if ti == 1 and model.share_weights:
            break
        for n in hf_parameter_order(tower.cfg, with_pooler=True):
            names.append((f"{prefix}.{n}", ti if n in tower.layout.entries else None, n))
    return [[e for e in names if not no_decay(e[0])], [e for e in names if no_decay(e[0])]]


class NwayTrainer:
    def __init__(self, model: NwayDualEncoder, *, loss: str = "lambda_mrr", T: float = 1.0, learning_rate: float = 7e-6,
                 weight_decay: float = 0.01, adam_epsilon: float = 1e-8, max_grad_norm: float = 1.0, warmup_steps: int = 4000,
                 total_steps: int = 100000, betas=(0.9, 0.999), bucket_layers: int = 1, reg_lambda: float = 0.0):
        if loss not in LOSS_KINDS:
            raise ValueError(f"loss must be one of {LOSS_KINDS}")
        self.model = model
        self.loss_kind, self.T = loss, T
        self.reg_lambda = float(reg_lambda)
        self.last_reg = None
        self.lr0, self.wd, self.eps, self.max_grad_norm = learning_rate, weight_decay, adam_epsilon, max_grad_norm
        self.warmup_steps, self.total_steps, self.betas = warmup_steps, total_steps, betas
        self.global_step = 0
        self.adam_step = 0          # bias-correction step of AdamW: differs from global_step only after resuming a reference fp16
                                    # checkpoint whose GradScaler skipped steps (state["step"] < global_step there)
        # CLDRD_FORCE_DDP=1: take the data-parallel path (constructor broadcast, bucket hooks, communication stream, Work handles) with a
        # process group of ONE rank too - the only way to run it over RCCL (ProcessGroupNCCL) on a one-GPU box (tests/test_gpu_ddp.py)
        self.distributed = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _env_flag("CLDRD_FORCE_DDP", "0") == "1")
        self.world = dist.get_world_size() if self.distributed else 1
        self.flat_p, self.flat_g = model.fuse_flat()
        # deferred weight gradients: one group launch per tower at the end of the backward on one GPU; with RCCL every
        # ceil(layers / 2) layers, so the first half of the buckets is all-reduced while the rest of the backward still runs
        for t in model.towers():
            t.wgrad_flush_layers = max(1, -(-t.cfg.n_layers // 2)) if self.distributed else 0
        dev = self.flat_p.device
        self._require_gpu(dev)
        n = self.flat_p.numel()
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        # weight-decay flag per 64-parameter chunk (every parameter starts on a 64 boundary)
        flags = torch.zeros(n // 64, dtype=torch.uint8)
        self.buckets = []          # (tower index, layer index or -1, start, end) in joint-flat coordinates
        for ti, (tower, toff) in enumerate(zip(model.towers(), model._tower_offsets)):
            for name in tower.layout.order:
                off, shape = tower.layout.entries[name]
                numel = 1
                for s in shape:
                    numel *= s
                prefix = "query_encoder." if ti == 0 else "passage_encoder."
                if not no_decay(prefix + name):
                    flags[(toff + off) // 64:(toff + off + numel + 63) // 64] |= 1
                if name.startswith("embeddings.") and name.endswith("_embeddings.weight"):
                    # bit 1: no bf16 / fp16 shadow (cldrd_adamw_step): the embedding kernels read the fp32 tables
                    flags[(toff + off) // 64:(toff + off + numel + 63) // 64] |= 2
            for li, (a, b) in enumerate(tower.layout.layer_range):
                self.buckets.append((ti, li, toff + a, toff + b))
            a, b = tower.layout.embed_range
            self.buckets.append((ti, -1, toff + a, toff + b))
        self.decay_flags = flags.to(dev)
        self.clip = torch.zeros(3, dtype=torch.float32, device=dev)
        # clip-norm partial sums: [0, half) the early piece (sqnorm_partial on the second stream), [half, 2 half) the late piece when it is taken by
        # sqnorm_partial, or [half, half + used) when the kernels that write the passage tower's layer gradients leave them (norm sink, round 5)
        # sink capacity from the layout: a weight-gradient group's slab reduction takes <= 256 slots per problem (4 per layer + the CLS-only
        # layer's split projections), a LayerNorm-parameter reduction ceil(3 d / 64) per LayerNorm (2 per layer + embeddings): 12-layer
        # towers (cfg4) need 13.4 k slots - a fixed 12 288 silently sent them back to the separate norm pass
        pt = model.towers()[-1].cfg
        self._sink_cap = 256 * (4 * pt.n_layers + 4) + ((3 * pt.dim + 63) // 64) * (2 * pt.n_layers + 2)
        self.norm_partial = torch.empty(ops.sqnorm_blocks() // 2 + self._sink_cap, dtype=torch.float32, device=dev)
        self.comm_stream = torch.cuda.Stream(device=dev) if self.distributed else None
        if self.comm_stream is not None:
            self.flat_g.record_stream(self.comm_stream)        # the bucket slices are used on it (the buffer lives as long as the trainer)
        # the query tower's stream, at the priority of torch's current stream (a high-priority stream changed nothing: profiles/r03_microbench.txt)
        self.q_stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        # third stream: scoring + loss + score backward between the towers' forward and backward (see forward_backward)
        self.l_stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self._pending = []
        self._sink_on = False
        # test hooks (tests/test_gpu_model.py, test_gpu_amp16.py prove the default path equal to the plain one they select; not configuration):
        self.zero_all_grads = False        # True: zero the whole gradient buffer and accumulate instead of writing every gradient once
        self.use_norm_sink = True          # False: the late clip-norm piece by a separate pass over the gradients
        self.window_schedule = False       # True: the query tower enqueued in slices released at the passage tower's LayerNorm / attention launches
                                           # (round 6, measured and NOT adopted: 11.58-11.92 ms per cfg2 step against 11.01 free-running, see forward_backward)
        # CLDRD_AMP=fp16 towers (encoder.py: amp16): the loss scale lives in device memory (hip_ops.new_loss_scale_state).  It is set every
        # step from dL/dCLS (ops.loss_scale_adapt: a power of two that puts the largest entering gradient at 2^11..2^12); what is kept of
        # torch.cuda.amp.GradScaler (the reference: nway_listwise_1.py:355-359) is its safety net: a non-finite gradient norm skips the step
        # and backs the scale off, `scale_growth_interval` (GradScaler's default 2000) finite steps give a factor 2 back.
        self.amp16 = any(getattr(t, "amp16", False) for t in model.towers())
        self.scale_growth_interval = 2000
        self._scale_state = ops.new_loss_scale_state(dev) if self.amp16 else None
        self._hyper = None          # device {lr, Adam step size} of the eager loss-scaled step (see _optimizer_launches)
        self._last_skip_h = False
        self._ddp_steps = 0         # train_step calls under torch.distributed (the captured-graph agreement happens at a fixed count)
        self._ddp_graph_key = None
        if self.distributed:
            # DDP constructor semantics (reference :250-255): rank 0's parameters win
            dist.broadcast(self.flat_p, src=0)
            dist.barrier()
        self._shadow = None
        self._joint_shadow()

    @staticmethod
    def _require_gpu(dev):
        if dev.type != "cuda":
            raise RuntimeError("NwayTrainer needs the model on a GPU (no CPU path)")

    # ---------------------------------------------------------------------------------------------------------
    def lr(self, step=None) -> float:
        step = self.global_step if step is None else step
        return self.lr0 * linear_schedule_factor(step, self.warmup_steps, self.total_steps)

    def _bucket_hook(self, ti):
        if not self.distributed:
            return None
        index = {(b[0], b[1]): (b[2], b[3]) for b in self.buckets}

        def hook(layer):
            # The bucket's gradients are complete on the stream the hook is called on: the communication stream waits for exactly
            # that point, then the all-reduce is issued ASYNCHRONOUSLY from it and its Work handle kept.  Nothing here relies on what a
            # blocking collective does to the caller's stream (ProcessGroupNCCL makes the current stream wait for its internal one,
            # gloo blocks the host): `_wait_pending` below is the one place where the result is ordered before its consumer.
            a, b = index[(ti, layer)]
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                work = dist.all_reduce(self.flat_g[a:b], op=dist.ReduceOp.SUM, async_op=True)
            self._pending.append(work)
        return hook

    def _early_norm_hook(self, main, side, defer=False):
        """One GPU: the clip norm is taken in two pieces.  When the passage tower's embedding block is complete (hook -1, called BEFORE
        its last weight-gradient group is launched) every gradient of the query tower (whose backward is already queued on the second
        stream) and of that block is final: [0, split) of the joint buffer, two thirds of it.  Their sums of squares are taken on the
        second stream UNDER the 3-ms weight-gradient launch (HBM reads next to an MFMA-bound kernel); after it only the passage
        tower's layers are left (`_optimizer_launches`).  Same fp64 reduction of fp32 partial sums as the one-piece norm."""
        toff = self.model._tower_offsets[-1]
        split = toff + self.model.towers()[-1].layout.embed_range[1]
        half = ops.sqnorm_blocks() // 2

        def hook(layer):
            if layer != -1 or split % 4 != 0 or split >= self.flat_g.numel():
                return
            ev = torch.cuda.Event()
            ev.record(main)
            if defer:
                # the query tower's backward is enqueued BEHIND the passage tower's (graph capture, see _backward): the partial norm has to
                # follow it on the second stream - `_finish_early_norm` - and still only waits for THIS point of the main stream
                self._norm_deferred = (ev, split, half)
                return
            side.wait_event(ev)
            with torch.cuda.stream(side):
                ops.sqnorm_partial(self.flat_g[:split], self.norm_partial, half)
            self._norm_split = split
        return hook

    # groups of the query tower's stepper (encoder.Stepper: one group = one GEMM / attention / LayerNorm launch, plus its split-K finish) released
    # per window of the passage tower: an attention kernel lasts ~55 us forward / ~107 us backward at cfg2, a LayerNorm kernel 29-50 us, a small
    # query-tower launch 7-10 us; one passage layer then releases seven groups = one query layer
    FWD_SLICE = {"attn": 3, "ln": 2}
    BWD_SLICE = {"attn": 3, "ln": 2}

    @staticmethod
    def _window(main, side, stepper, slices):
        def window(kind):
            if stepper.done:
                return
            ev = torch.cuda.Event()
            ev.record(main)                  # fires when the main stream reaches the HBM-bound kernel enqueued right after this call
            side.wait_event(ev)
            with torch.cuda.stream(side):
                stepper.step(slices[kind])
        return window

    def _finish_early_norm(self, side):
        d = getattr(self, "_norm_deferred", None)
        self._norm_deferred = None
        if d is None:
            return
        ev, split, half = d
        side.wait_event(ev)
        with torch.cuda.stream(side):
            ops.sqnorm_partial(self.flat_g[:split], self.norm_partial, half)
        self._norm_split = split

    def _wait_pending(self):
        """Order every outstanding bucket all-reduce before whatever the CURRENT stream runs next (the gradient norm / optimizer):
        ``Work.wait()`` makes the current stream wait for the collective under ProcessGroupNCCL (no host block) and blocks the host
        until completion under gloo; the stream-level join with the communication stream covers the copies gloo does on it."""
        pending, self._pending = self._pending, []
        for w in pending:
            w.wait()
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    def forward_backward(self, batch, _state_written=False):
        """Runs forward + backward (+ overlapped gradient all-reduce).  Returns (loss_out[2] device tensor, logits)."""
        model = self.model
        if self._state is not None and not _state_written and not torch.cuda.is_current_stream_capturing():
            # called on its own (tests, custom loops) while the towers read their dropout seeds from device memory: fresh seeds first
            seeds = [t.next_seed() for t in model.towers()]
            ops.write_step_state(self._state["seeds"], seeds[0], seeds[1] if len(seeds) > 1 else 0, None, 0.0, self.betas[0], self.betas[1], 1)
        qe, pe = model.query_encoder, model.passage_encoder
        q, nw = batch["query"], batch["nway_passages"]
        bz, nway, L = nw["input_ids"].shape
        write_once = not model.share_weights and not self.zero_all_grads      # (test hook: zero_all_grads = True zeroes + accumulates)
        main = torch.cuda.current_stream()
        # The query tower is ~1 % of the FLOPs but dozens of small, latency-bound launches: it runs on its own stream
        # next to the passage tower (forward and backward) instead of in front of it.
        side = self.q_stream if (not model.share_weights and self.q_stream is not None) else main
        if side is not main:
            side.wait_stream(main)
        # Step preamble that nothing in the passage tower's FORWARD depends on - on the second stream, next to that forward instead of in
        # front of it (134 us of a 12-ms step on the main stream until round 3, profiles/r03_microbench.txt): the zeros the embedding
        # tables' scatter-adds need and the transposed weight copies of the data-gradient GEMMs (left stale by the optimizer step).  The
        # main stream joins the second one after the forward, long before the backward reads either.
        with torch.cuda.stream(side):
            if not write_once:
                self.flat_g.zero_()                 # two tapes accumulate into one tower's gradients
            else:
                # every weight / bias / LayerNorm gradient is written exactly once per step (accumulate=False below); only the
                # embedding tables are scatter-added into and need zeros (2 x 94 MB instead of the whole 531 MB buffer)
                ops.zero_segments([self.flat_g[toff + t.layout.embed_range[0]:toff + t.layout.embed_range[1]]
                                   for t, toff in zip(model.towers(), model._tower_offsets)])      # one launch for both towers' tables
            for tower in model.towers():
                if not getattr(tower, "_t_fresh", False):
                    tower.refresh_transposed()
        # Round 6 experiment, OFF by default (`window_schedule`): the query tower enqueued in SLICES, each released by an event the main stream
        # records at the start of one of the passage tower's HBM-bound kernels (LayerNorm, attention), instead of running free next to everything -
        # the idea being that a 64 x 64-tile GEMM next to a kernel that waits for HBM costs nothing, while next to a large GEMM (a grid sized to
        # whole rounds of 256 CUs) it delays a round (0.32 ms of interference per step, profiles/r05_microbench.txt section 14).  Measured on one
        # box, interleaved rounds (profiles/r06_microbench.txt section 1): free-running 11.01 ms per step; slices of 3 / 2 groups per attention /
        # LayerNorm window 11.66; 2 / 1: 11.92; 4 / 3: 11.72; 7 / 7: 11.58.  The tighter the gating the slower: a gated chain of ~130 dependent
        # small launches finishes late (every release waits for the main stream to REACH a window, and the chain then still needs its own
        # 7-10 us per launch), so the join in front of the loss / the last weight-gradient group waits for it - that costs more than the
        # interference it avoids.  The stepping API (encoder.Stepper) stays: encode / backward are built on it.
        windows = side is not main and self.window_schedule
        self._windows_on = windows
        p_ids, p_mask = nw["input_ids"].reshape(bz * nway, L), nw["attention_mask"].reshape(bz * nway, L)
        if windows:
            with torch.cuda.stream(side):
                q_step = qe.encode_steps(q["input_ids"], q.get("attention_mask"), train=True, save=True, fp16=model.query_fp16, device_seed=True)
            p_cls, p_tape = pe.encode_steps(p_ids, p_mask, train=True, save=True, fp16=False, lengths=_lengths(nw), device_seed=True,
                                            window=self._window(main, side, q_step, self.FWD_SLICE)).finish()
            with torch.cuda.stream(side):
                q_cls, q_tape = q_step.finish()          # whatever is left (nothing, when the slices kept pace)
        else:
            with torch.cuda.stream(side):
                q_cls, q_tape = qe.encode(q["input_ids"], q.get("attention_mask"), train=True, save=True, fp16=model.query_fp16, device_seed=True)
            p_cls, p_tape = pe.encode(p_ids, p_mask, train=True, save=True, fp16=False, lengths=_lengths(nw), device_seed=True)      # "lengths" given: a packed batch
        if side is not main:
            main.wait_stream(side)
            q_cls.record_stream(main)
        mode = score_mode(model.in_batch_loss, model.all_in_batch_neg)
        Np = nway if mode == 0 else (bz * nway if mode == 1 else 2 * nway)
        # Scoring, loss and score backward on a stream of their OWN between the two joins.  In a replayed HIP graph a cross-stream edge is
        # honoured when the source stream's run of consecutive nodes ENDS, not at the node the edge leaves from: with the loss kernels on the
        # main stream the executor kept them and the query tower's backward (the branch captured first) in one run on one hardware queue, and
        # the passage tower's backward - the critical path - started only when that run was over: 0.62 ms of 5-10-us kernels with the rest of
        # the chip idle (kernel trace of the replayed step, profiles/r04_microbench.txt section 14).  A run that contains nothing but the loss
        # ends where both towers' backward begin.
        lst = self.l_stream if (side is not main and self.l_stream is not None) else main
        if lst is not main:
            lst.wait_stream(main)
        with torch.cuda.stream(lst):
            logits = torch.empty(bz, Np, dtype=torch.float32, device=q_cls.device)
            ops.score_fwd(q_cls, p_cls, logits, bz, nway, mode)
            labels = batch["labels"].to(device=logits.device, dtype=torch.float32)
            if mode != 0:   # in-batch negatives get the -0.5 label (reference nway_listwise_1.py:341-344)
                labels = torch.cat([labels, torch.full((bz, Np - nway), -0.5, dtype=torch.float32, device=logits.device)], dim=-1)
            loss_out, dlogits = ops.loss_fwd_bwd(self.loss_kind, logits, labels.contiguous(), T=self.T)
            if self.reg_lambda > 0.0 and mode == 0:     # reference nway_listwise_1.py:346-350: only without in-batch negatives
                self.last_reg = torch.empty(1, dtype=torch.float32, device=logits.device)
                ops.logit_norm_reg(logits, self.reg_lambda, loss_out, dlogits, self.last_reg)
            if self.world > 1:
                dlogits.mul_(1.0 / self.world)      # gradient mean over ranks == DDP's all-reduce / world_size
            dq, dp = torch.empty_like(q_cls), torch.empty_like(p_cls)
            ops.score_bwd(dlogits, q_cls, p_cls, dq, dp, bz, nway, mode)
            if self.amp16:
                ops.loss_scale_adapt(dq, dp, self._scale_state)          # dq, dp leave multiplied by S; every fp16 gradient downstream carries it
        if lst is not main:
            main.wait_stream(lst)
            for t_ in (logits, loss_out, dq, dp, q_cls, p_cls):
                t_.record_stream(main)
            q_cls.record_stream(lst)
            p_cls.record_stream(lst)
        with ops.loss_scale(self._scale_state.data_ptr() if self.amp16 else None, self.scale_growth_interval):
            return self._backward(batch, model, qe, pe, q_cls, p_cls, q_tape, p_tape, dlogits, dq, dp, bz, nway, mode, main, side, write_once, loss_out, logits,
                                  lst)

    def _backward(self, batch, model, qe, pe, q_cls, p_cls, q_tape, p_tape, dlogits, dq, dp, bz, nway, mode, main, side, write_once, loss_out, logits,
                  lst=None):
        """both towers' backward from dq / dp (scaled by the loss scale in the amp16 mode) + the bucket all-reduces"""
        if model.share_weights:
            # one tower, two tapes: gradients accumulate; all-reduce once everything is in
            pe.backward_from_cls(p_tape, dp)
            qe.backward_from_cls(q_tape, dq)
            if self.distributed:
                dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM)
        else:
            def query_backward():
                if side is not main:
                    side.wait_stream(lst if (lst is not None and lst is not main) else main)
                    dq.record_stream(side)
                with torch.cuda.stream(side):
                    qe.backward_from_cls(q_tape, dq, after_layer=self._bucket_hook(0), accumulate=not write_once)
            # The query tower's ~150 small backward launches run next to the passage tower's data-gradient chain.  (Next to its LAST
            # weight-gradient group instead - one ~3-ms launch off the critical path - measured +1.2 % step time, profiles/r03_microbench.txt:
            # that group loses more to the intruders than the GEMM chain does; the switch for it was removed in round 4.)
            # Order of the two towers' backward.  Eager: the query tower first - its ~150 small launches then run next to the passage tower's
            # data-gradient chain (enqueued behind ~150 passage launches they would start late and land on the last weight-gradient group:
            # +1.2 %, round 3).  Under GRAPH CAPTURE the order decides something else: the branch captured first behind the fork continues on
            # the hardware queue that ran the loss, the other one is released late - with the query tower first, the passage tower's backward
            # (the critical path) sat idle for 0.62 ms behind the loss in every replayed step (kernel trace, profiles/r04_microbench.txt
            # section 14).  So a captured step enqueues the passage tower first; the query tower's chain is released ~1.8 ms behind the loss
            # and still ends 4 ms before the join.
            windows = side is not main and getattr(self, "_windows_on", False)
            q_late = side is not main and (torch.cuda.is_current_stream_capturing() or windows)
            q_bstep = None
            if windows:
                # sliced like the forward: the query tower's backward is created here (inside the loss-scale context) and enqueued by the
                # passage tower's backward windows; what is left follows the passage tower's last weight-gradient group
                side.wait_stream(lst if (lst is not None and lst is not main) else main)
                dq.record_stream(side)
                with torch.cuda.stream(side):
                    q_bstep = qe.backward_steps(q_tape, dq, after_layer=self._bucket_hook(0), accumulate=not write_once)
            elif not q_late:
                query_backward()
            self._norm_split = None
            self._norm_deferred = None
            p_hook = self._bucket_hook(1)
            self._sink_on = False
            pe.norm_sink, pe.norm_sink_used = None, -1
            if p_hook is None and side is not main and write_once:
                p_hook = self._early_norm_hook(main, side, defer=q_late)
                if self.use_norm_sink:      # (test hook: False takes the late piece by a separate pass)
                    self._sink_on = True
                    pe.norm_sink = self.norm_partial[ops.sqnorm_blocks() // 2:ops.sqnorm_blocks() // 2 + self._sink_cap]
            if q_bstep is not None:
                pe.backward_steps(p_tape, dp, after_layer=p_hook, accumulate=not write_once,
                                  window=self._window(main, side, q_bstep, self.BWD_SLICE)).finish()
                with torch.cuda.stream(side):
                    q_bstep.finish()
                self._finish_early_norm(side)
            else:
                pe.backward_from_cls(p_tape, dp, after_layer=p_hook, accumulate=not write_once)
                if q_late:
                    # (released together with the passage tower's last weight-gradient group instead: +1.4 %, as in round 3's eager measurement)
                    query_backward()
                    self._finish_early_norm(side)
            main.wait_stream(side)
            if self.distributed:
                self._wait_pending()
        return loss_out, logits

    def optimizer_step(self):
        """clip_grad_norm_ + AdamW + scheduler.step of reference nway_listwise_1.py:355-367 (bf16: no GradScaler)."""
        self.global_step += 1
        self.adam_step += 1
        lr = self.lr(self.global_step - 1)          # the lr in effect during this step (scheduler steps afterwards)
        self._optimizer_launches(lr, self.adam_step)
        return lr

    def _optimizer_launches(self, lr, adam_step):
        """The device work of one optimizer step.  Under a captured step (`_state` installed) lr and the bias-corrected step size are
        read from device memory at replay time and the by-value arguments given here are ignored by the kernel."""
        with ops.loss_scale(self._scale_state.data_ptr() if self.amp16 else None, self.scale_growth_interval):     # clip_coef updates the scale
            self._norm_launches()
        towers = self.model.towers()
        hyper = self._state["hyper"] if self._state else None
        if hyper is None and self.amp16:
            # eager step of the loss-scaled mode: {lr, step size} go through device memory too, because the bias-correction exponent
            # is adam_step MINUS the steps the safety net skipped, and that count lives on the device (`_scale_state[3]`, updated by the
            # clip_coef launch just above; a skipped step leaves p / m / v untouched, so its own step size does not matter)
            if self._hyper is None:
                self._hyper = torch.zeros(2, dtype=torch.float32, device=self.flat_p.device)
            hyper = self._hyper
            ops.write_step_state(None, 0, 0, hyper, lr, self.betas[0], self.betas[1], adam_step, scale_state=self._scale_state)
        self._adamw_launches(lr, adam_step, towers, hyper)

    def _norm_launches(self):
        split = getattr(self, "_norm_split", None)
        if split:
            # [0, split) was summed on the second stream during the backward (`_early_norm_hook`; the main stream has joined it since).  The rest -
            # the passage tower's layer gradients - is final only behind its last weight-gradient group; when that group's slab reduction
            # and the LayerNorm-parameter reduction left their sums of squares (norm sink), nothing is re-read; else one more pass over them
            half = ops.sqnorm_blocks() // 2
            used = int(getattr(self.model.passage_encoder, "norm_sink_used", -1))
            if used > 0 and self._sink_on:
                ops.clip_coef(self.norm_partial, half + used, self.max_grad_norm, self.clip)
            else:
                if self._sink_on and used == -2 and not getattr(self, "_sink_warned", False):
                    self._sink_warned = True
                    import warnings
                    warnings.warn(f"clip norm: the norm sink ({self._sink_cap} slots) is too small for this tower's gradient launches; "
                                  "taking the late piece with a separate pass")
                ops.sqnorm_partial(self.flat_g[split:], self.norm_partial[half:2 * half], half)
                ops.clip_coef(self.norm_partial, 2 * half, self.max_grad_norm, self.clip)
            self._norm_split = None
        else:
            ops.grad_clip_coef(self.flat_g, self.max_grad_norm, self.norm_partial, self.clip)

    def _adamw_launches(self, lr, adam_step, towers, hyper=None):
        # one AdamW launch over the joint buffer; it also writes the bf16 shadow of every tower
        shadow = self._joint_shadow()
        # all-fp16 training: no pass of a training step reads the bf16 shadow (forward, backward and the transposed copies come from the fp16
        # one), so AdamW does not write it; the towers remember it is stale and an evaluation forward casts it first
        skip_h = self.amp16 and all(t.amp16 for t in towers)
        # ... and the fp16 shadow of the towers whose forward reads fp16 weights (the FFN GEMMs of every tower by default, the whole
        # high-precision pass of the query tower): one contiguous range of the joint buffer
        s16, r16 = self._joint_shadow16()
        fused16 = s16 is not None
        self._last_skip_h = bool(skip_h and fused16)
        with ops.optim_hyper(hyper.data_ptr() if hyper is not None else None):
            ops.adamw_step(self.flat_p, self.flat_g, self.m, self.v, self.decay_flags, None if (skip_h and fused16) else shadow, lr=lr, beta1=self.betas[0],
                           beta2=self.betas[1], eps=self.eps, weight_decay=self.wd, step=adam_step, clip=self.clip,
                           shadow16=s16[r16[0]:r16[1]] if fused16 else None, h16_range=r16 if fused16 else None)
        # the transposed copies wait for the next step's preamble on the second stream (forward_backward) when there is one
        defer_t = not self.model.share_weights and self.q_stream is not None
        for t in towers:
            t.refresh_shadows(need_transposed=not defer_t, cast=False, cast16=not fused16, h_stale=skip_h and fused16)

    def _joint_shadow(self):
        if getattr(self, "_shadow", None) is None:
            towers = self.model.towers()
            self._shadow = torch.empty(self.flat_p.numel(), dtype=torch.bfloat16, device=self.flat_p.device)
            self._joint_shadow16()
            for t, off in zip(towers, self.model._tower_offsets):
                t.flat_h = self._shadow[off:off + t.layout.total]
                t.refresh_shadows(need_transposed=True)
        return self._shadow

    def _joint_shadow16(self):
        """(joint fp16 shadow or None, (begin, end) of the parameters it mirrors): the towers that need fp16 weights are adjacent in the
        joint buffer, so one range covers them; each such tower's ``flat_h16`` is a slice of the joint tensor."""
        if getattr(self, "_shadow16", None) is None:
            towers, offs = self.model.towers(), self.model._tower_offsets
            need = [(t, off) for t, off in zip(towers, offs) if t.needs_h16]
            if not need:
                self._shadow16 = (None, None)
            else:
                lo, hi = need[0][1], need[-1][1] + need[-1][0].layout.total
                if sum(t.layout.total for t, _ in need) != hi - lo:
                    raise RuntimeError("towers with fp16 weights are not adjacent in the joint buffer")
                buf = torch.empty(self.flat_p.numel(), dtype=torch.float16, device=self.flat_p.device)
                for t, off in need:
                    t.flat_h16 = buf[off:off + t.layout.total]
                    t._shadow_version = -1
                self._shadow16 = (buf, (lo, hi))
        return self._shadow16

    def train_step(self, batch):
        """One full step; returns the device tensor {loss, pair count} (no host sync).

        After ``CLDRD_GRAPH_WARMUP`` (3) eager steps on one batch shape the step - forward, loss, backward, clip, AdamW, shadows, both
        streams: ~370 kernel launches - is captured into a HIP graph and replayed (``CLDRD_GRAPH=0``: always eager; never under
        torch.distributed: the bucket all-reduces are issued from hooks).  The host then spends ~0.1 ms per step instead of ~3.7 ms,
        which is what an enqueue-bound shape (cfg1: B = 4, N = 8) and eight rank processes sharing one host need.  What changes
        from step to step - the two towers' dropout seeds, lr, Adam's bias-corrected step size - lives in device memory and is
        written by one small launch in front of each replay (``ops.write_step_state``); results are bit-identical to the eager path."""
        self._joint_shadow()
        nw = batch["nway_passages"]
        lens = _lengths(nw)
        if lens is not None:
            bz_, nway_, L_ = nw["input_ids"].shape
            if not self.model.passage_encoder.would_pack(lens, bz_ * nway_, L_, has_mask=nw.get("attention_mask") is not None):
                # "lengths" came with the batch (batch_to_device attaches them to every right-padded batch) but the encoder will not pack it
                # (fill above 92 %, CLDRD_PACK=0): it is an ordinary padded batch and takes the graph path like one
                batch = dict(batch.items())
                batch["nway_passages"] = {k: v for k, v in nw.items() if k != "lengths"}
                lens = None
        # Data-parallel ranks that may replay captured collectives agree on it ONCE, at call number _DDP_WARM + 1 of EVERY rank: the count
        # advances here, in front of the per-batch gates below (whether THIS rank's batch is packed depends on its own token fill, so a
        # count taken behind the gate would reach the agreement step at different global steps on different ranks, or never: the one-off
        # all-reduce would then meet another rank's bucket all-reduces - a hang).  A rank whose batch is not eligible at that step votes "no".
        ddp_graph = self.distributed and self._graph_wanted()
        if ddp_graph:
            self._ddp_steps += 1
        if self._graph_wanted() and lens is None:      # a packed batch changes its row count every step: eager
            out = self._train_step_graph(batch)
            if out is not None:
                return out
        elif ddp_graph and self._ddp_steps == self._DDP_WARM + 1:
            self._agree_on_capture(False)              # packed batch at the agreement step: every rank stays eager for good
        if self._state is not None:
            # a captured shape exists, so the towers read their seeds (and AdamW its lr) from device memory: an eager step (another
            # batch shape) writes this step's values there first, exactly as a replay does
            lr = self._advance_step_state()
            loss_out, logits = self.forward_backward(batch, _state_written=True)
            self._optimizer_launches(lr, self.adam_step)
        else:
            loss_out, logits = self.forward_backward(batch)
            self.optimizer_step()
        self.last_logits = logits
        return loss_out

    def _advance_step_state(self):
        """Host bookkeeping of one step (what optimizer_step / encode do in the eager path) + the launch that writes the step's seeds, lr
        and Adam step size to device memory, in stream order in front of the launches that read them.  Returns lr."""
        self.global_step += 1
        self.adam_step += 1
        lr = self.lr(self.global_step - 1)
        seeds = [t.next_seed() for t in self.model.towers()]
        ops.write_step_state(self._state["seeds"], seeds[0], seeds[1] if len(seeds) > 1 else 0, self._state["hyper"], lr, self.betas[0],
                             self.betas[1], self.adam_step, scale_state=self._scale_state)
        return lr

    def skipped_steps(self) -> int:
        """Steps the loss-scale safety net skipped so far (a device -> host read: logging / checkpoint time only)."""
        return int(self._scale_state[3].item()) if self._scale_state is not None else 0

    # ---- HIP-graph replay of the step --------------------------------------------------------------------------------------
    _state = None
    _DDP_WARM = 3               # eager steps in front of the capture (and, under torch.distributed, in front of the agreement step)

    def _graph_wanted(self):
        if self.distributed:
            # data-parallel ranks: the bucket all-reduces are issued from hooks INSIDE the step.  Over ProcessGroupNCCL (= RCCL) they can be
            # captured with it (the collective's launch is recorded on its stream, Work.wait() becomes an event edge of the graph); gloo
            # (tests) copies through the host and cannot.  With MORE THAN ONE rank this is opt-in (CLDRD_DDP_GRAPH=1): capture + replay of
            # collectives has only ever run with one rank on this pool (one GPU per box), so real multi-rank jobs launch eagerly by default;
            # a one-rank process group (CLDRD_FORCE_DDP: the bench's ddp leg, tests) captures by default.
            if _env_flag("CLDRD_DDP_GRAPH", "1" if self.world == 1 else "0") != "1" or dist.get_backend() != "nccl":
                return False
        return (not self.model.share_weights and not getattr(self, "_graph_broken", False)
                and _env_flag("CLDRD_GRAPH", "1") != "0" and self.flat_p.is_cuda)

    @staticmethod
    def _batch_key(batch):
        q, nw = batch["query"], batch["nway_passages"]
        return (tuple(q["input_ids"].shape), q.get("attention_mask") is not None, tuple(nw["input_ids"].shape), tuple(batch["labels"].shape))

    def _train_step_graph(self, batch):
        key = self._batch_key(batch)
        graphs = self.__dict__.setdefault("_graphs", {})
        entry = graphs.get(key)
        if entry is None:
            entry = graphs[key] = {"seen": 0, "graph": None}
        warm = self._DDP_WARM
        ddp = self.distributed        # (a one-rank forced group, CLDRD_FORCE_DDP, takes the same path: the agreement is then a one-rank all-reduce)
        if ddp:
            # Ranks that replay captured collectives must all do so for the SAME steps, or the collectives of one rank's replay meet another
            # rank's eager launches in another order / never (a hang no capture-time check can see).  So: (a) the decision is taken ONCE, at
            # train_step call number `warm` + 1 on every rank (a count, not a per-shape condition: ranks with dynamic padding see different
            # shapes); there each rank tries to capture its current batch shape if it has seen nothing else so far, and the ranks agree with
            # one all_reduce(MIN) of the outcome: a single failure anywhere leaves every rank eager for good; (b) afterwards only that ONE
            # shape replays - an eager step of another shape issues the same bucket all-reduces in the same order as a replay does, so
            # mixing the two across ranks is safe, a second capture (another agreement collective at a rank-dependent step) would not be.
            # (`_ddp_steps` is advanced by train_step for every call, eligible for the graph path or not)
            if self._ddp_steps <= warm:
                entry["seen"] += 1
                return None
            if self._ddp_steps > warm + 1:
                if entry["graph"] is None:
                    return None
            elif len(graphs) > 1 or entry["seen"] < warm:
                self._agree_on_capture(False)
                return None
        if entry["graph"] is None and entry["seen"] < warm:
            entry["seen"] += 1              # eager: allocator warm-up, one-time kernel attributes
            return None
        dev = self.flat_p.device
        if self._state is None:
            self._state = {"seeds": torch.zeros(2, dtype=torch.int64, device=dev), "hyper": torch.zeros(2, dtype=torch.float32, device=dev)}
        towers = self.model.towers()
        if entry["graph"] is None:
            if len(graphs) > 4:            # a few shapes at most (the last partial batch is dropped: drop_last): anything else stays eager
                return None
            ok = True
            try:
                entry.update(self._capture(batch))
                entry["h_stale"] = self._last_skip_h        # the captured AdamW does not write the bf16 shadow (see the replay below)
            except Exception as exc:       # a capture problem must never take training down: fall back to the eager step for good
                import warnings
                first = exc
                while first.__context__ is not None:          # what went wrong INSIDE the capture (the capture's own exit error hides it)
                    first = first.__context__
                warnings.warn(f"HIP-graph capture of the training step failed ({type(first).__name__}: {first}); staying eager")
                ok = False
            if ddp:
                ok = self._agree_on_capture(ok)
                if not ok:
                    entry["graph"] = None
            if not ok:
                self._graph_broken = True
                for t in towers:
                    t.seed_base_ptr = None
                    t._t_fresh = False       # refresh_transposed() inside the failed capture set the flag without executing a kernel
                self._state = None
                self._norm_split = None
                self._norm_deferred = None
                torch.cuda.synchronize()
                return None
        # this step's inputs and state, in stream order in front of the replay
        dsts, srcs = entry["inputs"], self._flat_inputs(batch)
        if all(s_.is_cuda and s_.is_contiguous() and s_.dtype == d_.dtype and s_.shape == d_.shape for d_, s_ in zip(dsts, srcs)):
            ops.copy_segments(dsts, srcs)           # one launch instead of five
        else:
            for dst, src in zip(dsts, srcs):
                dst.copy_(src, non_blocking=True)
        self._advance_step_state()
        entry["graph"].replay()
        if entry.get("h_stale"):
            # The captured AdamW skipped the bf16 weight shadow (all-fp16 training reads none of it).  The Python call that records this
            # (refresh_shadows(h_stale=True)) ran at capture time only; a replay changes the weights again, so every replay has to leave the
            # towers marked - or an evaluation forward after further replays would encode with the bf16 matrices of the LAST evaluation.
            for t in towers:
                t._h_stale = True
        self.last_logits = entry["logits"]
        return entry["loss_out"]

    def _agree_on_capture(self, ok: bool) -> bool:
        """MIN over ranks of "my capture worked" (one small all-reduce + a host read, once per job)."""
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.flat_p.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        agreed = bool(int(flag.item()))
        if not agreed and ok:
            import warnings
            warnings.warn("HIP-graph capture of the training step failed on another rank; every rank stays eager")
        if not agreed:
            self._graph_broken = True
        return agreed

    @staticmethod
    def _flat_inputs(batch):
        q, nw = batch["query"], batch["nway_passages"]
        t = [q["input_ids"], nw["input_ids"], nw["attention_mask"], batch["labels"]]
        if q.get("attention_mask") is not None:
            t.append(q["attention_mask"])
        return t

    def _capture(self, batch):
        dev = self.flat_p.device
        towers = self.model.towers()
        q, nw = batch["query"], batch["nway_passages"]
        static = {"query": {"input_ids": q["input_ids"].to(dev).clone()},
                  "nway_passages": {"input_ids": nw["input_ids"].to(dev).clone(), "attention_mask": nw["attention_mask"].to(dev).clone()},
                  "labels": batch["labels"].to(device=dev, dtype=torch.float32).clone()}
        if q.get("attention_mask") is not None:
            static["query"]["attention_mask"] = q["attention_mask"].to(dev).clone()
        for i, t in enumerate(towers):
            t.seed_base_ptr = self._state["seeds"].data_ptr() + 8 * i
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        # the captured launches take seeds / lr from device memory: nothing of THIS step's values is baked in (the counters are not
        # advanced here; the capture itself executes nothing)
        # thread_local: only THIS thread's calls are checked against the capture.  The default ("global") makes any HIP call of any other thread
        # an error while the capture runs - and a process with a ProcessGroupNCCL has such a thread: the watchdog polls the events of earlier
        # collectives (seen as a 1-in-3 failure of the one-rank RCCL test when the whole suite ran before it).
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            try:
                loss_out, logits = self.forward_backward(static)
                self._optimizer_launches(0.0, 1)
            except BaseException:
                # join every stream forked inside the capture before it ends: otherwise ending it fails too ("unjoined work"), the stream
                # stays in capture mode and the eager fallback cannot run either
                cur = torch.cuda.current_stream()
                for s_ in (self.q_stream, self.l_stream, self.comm_stream):
                    if s_ is not None:
                        cur.wait_stream(s_)
                raise
        return {"graph": g, "inputs": self._flat_inputs(static), "loss_out": loss_out, "logits": logits, "static": static}

    # ---------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def train_metrics(self, logits, labels, topk: int = 10):
        """Per-batch train MRR@k / Recall@k w.r.t. the ``labels == 1`` position (reference :375-385); call on logging steps."""
        order = torch.argsort(logits.float(), dim=-1, descending=True, stable=True)
        lab = torch.gather(labels.to(logits.device).float(), 1, order).cpu().numpy()
        import numpy as np
        first = np.where(lab == 1)[1]
        keep = first[first < topk]
        if len(keep) == 0:
            return 0.0, 0.0
        return float(np.sum(1.0 / (keep + 1.0)) / len(first)), float(len(keep) / len(first))

    # ---- checkpoint payload (reference nway_listwise_1.py:418-426 / :300-304) --------------------------------------------
    def _optimizer_names(self):
        return optimizer_param_groups(self.model)

    def _slice(self, buf, ti, hf_name):
        tower, toff = self.model.towers()[ti], self.model._tower_offsets[ti]
        off, shape = tower.layout.entries[hf_name]
        numel = 1
        for d in shape:
            numel *= d
        return buf[toff + off:toff + off + numel].view(shape)

    def optimizer_state_dict(self):
        """``torch.optim.Optimizer.state_dict()`` layout of the reference's AdamW (state: step / exp_avg / exp_avg_sq per parameter
        index; two param groups), so a reference run can resume from a checkpoint written here and vice versa."""
        groups = self._optimizer_names()
        state, param_groups, idx = {}, [], 0
        applied = max(0, self.adam_step - self.skipped_steps())      # scaler.step() semantics: skipped steps do not count (reference :357)
        for gi, entries in enumerate(groups):
            ids = []
            for _, ti, n in entries:
                if ti is not None and applied > 0:
                    state[idx] = {"step": applied, "exp_avg": self._slice(self.m, ti, n).detach().cpu().clone(),
                                  "exp_avg_sq": self._slice(self.v, ti, n).detach().cpu().clone()}
                ids.append(idx)
                idx += 1
            param_groups.append({"weight_decay": self.wd if gi == 0 else 0.0, "lr": self.lr(), "initial_lr": self.lr0,
                                 "betas": tuple(self.betas), "eps": self.eps, "correct_bias": True, "params": ids})
        return {"state": state, "param_groups": param_groups}

    def load_optimizer_state_dict(self, sd):
        """Accepts the torch layout above (a reference checkpoint's ``optimizer``) or this package's round-1 flat ``{"m","v"}``
        blob; anything else raises (the reference's ``optimizer.load_state_dict`` would, :302)."""
        if not isinstance(sd, dict):
            raise ValueError(f"optimizer state must be a dict, got {type(sd).__name__}")
        if "m" in sd and "v" in sd:
            if sd["m"].numel() != self.m.numel() or sd["v"].numel() != self.v.numel():
                raise ValueError("optimizer state: flat m/v size does not match this model")
            self.m.copy_(sd["m"])
            self.v.copy_(sd["v"])
            self._opt_step_loaded = int(sd["step"]) if "step" in sd else None
            return
        if "state" not in sd or "param_groups" not in sd:
            raise ValueError(f"optimizer state: unknown layout (keys {sorted(sd)[:6]})")
        groups = self._optimizer_names()
        pg = sd["param_groups"]
        if len(pg) != 2 or any(len(g["params"]) != len(e) for g, e in zip(pg, groups)):
            raise ValueError("optimizer state: parameter groups do not match this model "
                             f"(expected sizes {[len(e) for e in groups]}, got {[len(g['params']) for g in pg]})")
        self.m.zero_()
        self.v.zero_()
        steps = set()
        index = {}
        for g, entries in zip(pg, groups):
            for pid, e in zip(g["params"], entries):
                index[pid] = e
        for pid, st in sd["state"].items():
            if pid not in index:
                raise ValueError(f"optimizer state: unknown parameter index {pid}")
            name, ti, n = index[pid]
            if ti is None:
                continue                                  # BERT pooler: never has a gradient on this path
            for key, buf in (("exp_avg", self.m), ("exp_avg_sq", self.v)):
                t = st[key]
                dst = self._slice(buf, ti, n)
                if tuple(t.shape) != tuple(dst.shape):
                    raise ValueError(f"optimizer state: {name}.{key} has shape {tuple(t.shape)}, expected {tuple(dst.shape)}")
                dst.copy_(t.to(torch.float32))
            steps.add(int(st["step"]))
        if len(steps) > 1:
            raise ValueError(f"optimizer state: per-parameter step counts differ ({sorted(steps)[:4]}...): one fused step counter here")
        self._opt_step_loaded = steps.pop() if steps else None

    def state_dict(self):
        """Checkpoint payload in the reference's shape (nway_listwise_1.py:418-426): DDP-style ``module.`` prefixed keys, the
        optimizer in torch's ``state_dict()`` layout, the scheduler as ``LambdaLR.state_dict()`` would give it."""
        return {"global_step": self.global_step,
                "state_dict": {"module." + k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()},
                "optimizer": self.optimizer_state_dict(),
                "scheduler": {"base_lrs": [self.lr0, self.lr0], "last_epoch": self.global_step, "_step_count": self.global_step + 1,
                              "_get_lr_called_within_step": False, "_last_lr": [self.lr(), self.lr()], "lr_lambdas": [None, None],
                              "warmup_steps": self.warmup_steps, "total_steps": self.total_steps}}

    def load_state_dict(self, ckpt):
        sd = {k[7:] if k.startswith("module.") else k: v for k, v in ckpt["state_dict"].items()}
        self.model.load_state_dict(sd)
        self._opt_step_loaded = None
        if "optimizer" in ckpt:
            self.load_optimizer_state_dict(ckpt["optimizer"])
        self.global_step = int(ckpt.get("global_step", 0))
        if "scheduler" in ckpt and isinstance(ckpt["scheduler"], dict) and "last_epoch" in ckpt["scheduler"] and "global_step" not in ckpt:
            self.global_step = int(ckpt["scheduler"]["last_epoch"])
        self.adam_step = self.global_step if self._opt_step_loaded is None else self._opt_step_loaded
        if self._scale_state is not None:
            self._scale_state[3] = 0.0          # the loaded step already excludes the skipped ones
        for t in self.model.towers():
            t.refresh_shadows(need_transposed=True)


# =================================================================================================================
# Command line of reference trainer/multistep-curriculum/nway_listwise_{1,2,3}.py (flags, defaults, log and checkpoint
# formats), plus: --loss (the reference hard-codes lambda_mrr), --token_cache_dir (tokenise once) and
# --synthetic_steps (run on generated MSMARCO-shaped batches when no dataset / tokenizer is at hand).
# =================================================================================================================
def get_args(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="N-way listwise distillation training on MI355X (CL-DRD multistep curriculum)")
    ap.add_argument("--queries_path", default=None)
    ap.add_argument("--collection_path", default=None)
    ap.add_argument("--training_path", default=None)
    ap.add_argument("--experiment_folder", default="experiments/multistep-curriculum/")
    ap.add_argument("--model_name_or_path", default="sebastian-hofstaetter/distilbert-dot-tas_b-b256-msmarco")
    ap.add_argument("--tokenizer_name_or_path", default="distilbert-base-uncased")
    ap.add_argument("--resume", default=None)
    ap.add_argument("--model_checkpoint", default=None)
    ap.add_argument("--seed", default=4680, type=int)
    ap.add_argument("--show_progress", default=True, type=bool)
    ap.add_argument("--run_folder", default="experiment")
    ap.add_argument("--log_dir", default="log/")
    ap.add_argument("--logging_steps", default=50, type=int)
    ap.add_argument("--evaluate_steps", default=10000, type=int)
    ap.add_argument("--model_save_dir", default="models")
    ap.add_argument("--learning_rate", default=7e-6, type=float)
    ap.add_argument("--weight_decay", default=0.01, type=float)
    ap.add_argument("--adam_epsilon", default=1e-8, type=float)
    ap.add_argument("--max_grad_norm", default=1.0, type=float)
    ap.add_argument("--num_train_epochs", default=4, type=int)
    ap.add_argument("--warmup_steps", default=4000, type=int)
    ap.add_argument("--reg_lambda", default=0.0, type=float)
    ap.add_argument("--query_max_len", default=30, type=int)
    ap.add_argument("--passage_max_len", default=256, type=int)
    ap.add_argument("--use_fp16", default=True, type=bool, help="accepted for compatibility: MFMA operands are fp16 (loss-scaled backward) with fp32 master weights; CLDRD_AMP=bf16 for a bf16 backward")
    ap.add_argument("--train_batch_size", default=8, type=int)
    ap.add_argument("--share_weights", action="store_true", default=False)
    ap.add_argument("--label_mode", default="8", type=str)
    ap.add_argument("--in_batch_loss", action="store_true", default=False)
    ap.add_argument("--all_in_batch_neg", action="store_true", default=False)
    ap.add_argument("--n_gpu", default=1, type=int)
    ap.add_argument("--local_rank", default=-1, type=int)
    ap.add_argument("--loss", default="lambda_mrr", choices=LOSS_KINDS)
    ap.add_argument("--token_cache_dir", default=None)
    ap.add_argument("--loader_workers", default=4, type=int, help="collate (tokenise / gather from the token cache) worker processes; "
                    "batches arrive in pinned memory, a few steps ahead of the GPU")
    ap.add_argument("--synthetic_steps", default=0, type=int, help="steps per epoch on generated batches (no dataset files needed)")
    ap.add_argument("--synthetic_nway", default=30, type=int)
    ap.add_argument("--synthetic_fixed", action="store_true", default=False, help="every synthetic passage passage_max_len tokens long (default: MS MARCO-shaped lengths)")
    ap.add_argument("--synthetic_model", default="distilbert", choices=("distilbert", "tiny"),
                    help="random-init architecture for --synthetic_steps runs without a model directory (tiny: encoder.tiny_config)")
    args = ap.parse_args(argv)
    args.run_folder = os.path.join(args.experiment_folder, args.run_folder)
    args.log_dir = os.path.join(args.run_folder, args.log_dir)
    args.model_save_dir = os.path.join(args.run_folder, args.model_save_dir)
    return args


def set_env(args):
    """One process per GPU: ranks come from torchrun's environment (or --local_rank as in the reference, :40-49)."""
    if args.local_rank == -1 and "LOCAL_RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        args.local_rank = int(os.environ["LOCAL_RANK"])
    if args.local_rank != -1:
        torch.cuda.set_device(args.local_rank)
        if not dist.is_initialized():
            dist.init_process_group("nccl")
        args.nranks = dist.get_world_size()
        args.distributed = args.nranks > 1
    else:
        args.nranks, args.distributed = 1, False
    args.device = torch.device("cuda", max(args.local_rank, 0))
    args.rank = dist.get_rank() if args.distributed else 0
    cap_host_threads()
    return args


def write_train_logs(epoch, step, loss_val, mrr_val, recall_val, lr, filename, cutoff=10, **kwargs):
    """Tab-separated log of reference :78-90 (including its quirk: the first call only writes the header)."""
    if not os.path.exists(filename):
        with open(filename, "w") as fh:
            fh.write("\t".join(["epoch", "step", "loss_val", f"mrr@{cutoff}", f"recall@{cutoff}", "lr"] + list(kwargs)) + "\n")
    else:
        with open(filename, "a") as fh:
            fh.write(f"{epoch}\t{step}\t{loss_val:.3f}\t{mrr_val:.3f}\t{recall_val:.3f}\t{lr:.10f}")
            for v in kwargs.values():
                fh.write(f"\t{v:.3f}")
            fh.write("\n")


class _Avg:
    def __init__(self):
        self.reset()

    def reset(self):
        self.sum, self.n = 0.0, 0

    def update(self, v):
        self.sum += float(v)
        self.n += 1

    @property
    def avg(self):
        return self.sum / max(self.n, 1)


def build_dataloader(args):
    """Label mode -> file constructor as reference :173-245; per-rank batch = train_batch_size // nranks, shuffle, drop_last."""
    from transformers import AutoTokenizer
    from ..dataset.nway_dataset import NwayDataset
    tok = AutoTokenizer.from_pretrained(args.tokenizer_name_or_path)
    a = (args.queries_path, args.collection_path, args.training_path, tok)
    kw = dict(max_query_len=args.query_max_len, max_passage_len=args.passage_max_len, label_mode=args.label_mode)
    shard = dict(rank=args.rank, nranks=args.nranks) if args.distributed else {}
    mode = args.label_mode
    if mode == "1":
        if args.distributed:
            raise NotImplementedError
        ds = NwayDataset.create_from_json_line_file(*a, **kw)
    elif mode in ("2", "4"):
        ds = NwayDataset.create_from_relT_most_semi_hard_file(*a, **kw, **shard)
    elif mode in ("3", "9"):
        ds = NwayDataset.create_from_10relT_20neg_file(*a, **kw, **shard)
    elif mode in ("5", "10"):
        ds = NwayDataset.create_from_20relT_10neg_file(*a, **kw, **shard)
    elif mode == "6":
        ds = NwayDataset.create_from_30relT_file(*a, **kw, **shard)
    elif mode in ("7", "8"):
        ds = NwayDataset.create_from_5relT_25neg_file(*a, **kw, **shard)
    else:
        raise ValueError(f"label mode {mode} not implemented")
    if args.token_cache_dir:
        # every rank holds the full query / passage tables, so the cache is the same everywhere: rank 0 builds and writes it
        # (atomic renames), the others load it after the barrier instead of racing on the same files
        if args.distributed:
            if args.rank == 0:
                ds.with_token_cache(args.token_cache_dir)
            dist.barrier()
            if args.rank != 0:
                ds.with_token_cache(args.token_cache_dir, build=False)
        else:
            ds.with_token_cache(args.token_cache_dir)
    assert args.train_batch_size % args.nranks == 0
    g = torch.Generator()
    g.manual_seed(args.seed + args.rank)
    nw = max(int(getattr(args, "loader_workers", 1)), 0)
    return ds, torch.utils.data.DataLoader(ds, batch_size=args.train_batch_size // args.nranks, shuffle=True, num_workers=nw,
                                           collate_fn=ds.collate_fn, drop_last=True, generator=g, pin_memory=True,
                                           **(dict(prefetch_factor=4, persistent_workers=True) if nw else {}))


class _SyntheticBatches(torch.utils.data.Dataset):
    """``--synthetic_steps`` batches of the collate_fn layout from the portable generator (synthetic.nway_batch): item i IS batch i (a
    function of (seed, rank, i) only), so the loader below hands them out with ``batch_size=None`` from worker processes, pinned, like the
    real one."""

    def __init__(self, args):
        self.args = args

    def __len__(self):
        return self.args.synthetic_steps

    def __getitem__(self, i):
        from .. import synthetic as syn
        a = self.args
        b = syn.nway_batch(a.seed + 1000 * a.rank + i, a.train_batch_size // a.nranks, a.synthetic_nway, a.query_max_len,
                           a.passage_max_len, vocab=getattr(a, "synthetic_vocab", syn.VOCAB), ragged=not getattr(a, "synthetic_fixed", False),
                           label_kind="teacher" if a.loss in ("kl_div", "margin_mse") else "mode9")
        return attach_lengths(b)


def _synthetic_loader(args):
    nw = max(int(getattr(args, "loader_workers", 1)), 0)
    return torch.utils.data.DataLoader(_SyntheticBatches(args), batch_size=None, shuffle=False, num_workers=nw, pin_memory=True,
                                       **(dict(prefetch_factor=4, persistent_workers=True) if nw else {}))


def common_steps_per_epoch(local_steps: int, distributed: bool, dev=None, group=None) -> int:
    """Examples are sharded ``line_idx % nranks`` with ``drop_last`` (reference dataset/nway_dataset.py:305), so ranks can differ by
    one batch per epoch; every rank runs MIN over ranks steps: same schedule length (t_total) everywhere and no rank left alone in
    a gradient all-reduce at the end of an epoch."""
    if not distributed:
        return int(local_steps)
    t = torch.tensor([int(local_steps)], dtype=torch.int64, device=dev if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return int(t.item())


def batch_to_device(batch, dev):
    """Move a collated batch to the device (asynchronously when the loader pinned it); "lengths" stay on the host."""
    batch = attach_lengths(batch)
    out = {}
    for k, v in batch.items():
        if isinstance(v, torch.Tensor):
            out[k] = v.to(dev, non_blocking=True)
        elif hasattr(v, "items"):
            out[k] = {kk: (vv.to(dev, non_blocking=True) if (isinstance(vv, torch.Tensor) and kk != "lengths") else vv) for kk, vv in v.items()}
        else:
            out[k] = v
    return out


def train(args):
    """Epoch / step loop of reference :328-426 on top of NwayTrainer.train_step."""
    import random
    import numpy as np
    dev = args.device
    if args.synthetic_steps > 0:
        loader, n_examples = _synthetic_loader(args), args.synthetic_steps * args.train_batch_size
    else:
        ds, loader = build_dataloader(args)
        n_examples = len(ds)
    if args.synthetic_steps > 0 and not os.path.isdir(str(args.model_name_or_path)):
        from ..encoder import EncoderConfig
        if args.synthetic_model == "tiny":
            from ..encoder import tiny_config
            cfg = tiny_config()
        else:
            cfg = EncoderConfig(arch="distilbert")
        args.synthetic_vocab = cfg.vocab_size
        model = NwayDualEncoder(cfg, share_weights=args.share_weights, in_batch_loss=args.in_batch_loss,
                                all_in_batch_neg=args.all_in_batch_neg)
    else:
        model = NwayDualEncoder(args.model_name_or_path, share_weights=args.share_weights, in_batch_loss=args.in_batch_loss,
                                all_in_batch_neg=args.all_in_batch_neg)
    model.to(dev)
    model.train()
    steps_per_epoch = common_steps_per_epoch(len(loader), args.distributed, dev)
    t_total = steps_per_epoch * args.num_train_epochs
    trainer = NwayTrainer(model, loss=args.loss, learning_rate=args.learning_rate, weight_decay=args.weight_decay,
                          adam_epsilon=args.adam_epsilon, max_grad_norm=args.max_grad_norm, warmup_steps=args.warmup_steps,
                          total_steps=t_total, reg_lambda=args.reg_lambda)
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)            # after model / loader construction, as the reference does (:282)
    start_epoch = 0
    if args.resume:
        assert args.model_checkpoint is None
        ckpt = torch.load(args.resume, map_location="cpu", weights_only=False)
        trainer.load_state_dict(ckpt)
        start_epoch = ckpt["epoch"] - 1
    if args.model_checkpoint:
        assert args.resume is None
        ckpt = torch.load(args.model_checkpoint, map_location="cpu", weights_only=False)
        trainer.load_state_dict({"state_dict": ckpt["state_dict"]})
    main_rank = args.rank == 0
    if main_rank:
        os.makedirs(args.log_dir, exist_ok=True)
        os.makedirs(args.model_save_dir, exist_ok=True)
    loss_m, mrr_m, rec_m, reg_m, ratio_m = _Avg(), _Avg(), _Avg(), _Avg(), _Avg()
    topk = 10
    log_file = os.path.join(args.log_dir, "train_logs.log")
    for epoch in range(start_epoch, args.num_train_epochs):
        for step_i, batch in enumerate(loader):
            if step_i >= steps_per_epoch:       # ranks whose shard is one batch longer stop with the others (no unmatched all-reduce)
                break
            batch = batch_to_device(batch, dev)
            loss_out = trainer.train_step(batch)
            if main_rank and trainer.global_step % args.logging_steps == 0:
                # the reference reads loss / MRR back every step (:369-389); here only on logging steps (no per-step sync)
                labels = batch["labels"]
                logits = trainer.last_logits
                if logits.shape[1] != labels.shape[1]:
                    labels = torch.cat([labels, torch.full((labels.shape[0], logits.shape[1] - labels.shape[1]), -0.5,
                                                           device=labels.device)], dim=-1)
                b_mrr, b_rec = trainer.train_metrics(logits, labels, topk)
                loss_val = float(loss_out[0].item())
                loss_m.update(loss_val), mrr_m.update(b_mrr), rec_m.update(b_rec)
                extra = {}
                if args.reg_lambda > 0.0 and trainer.last_reg is not None:
                    reg = float(trainer.last_reg.item())
                    reg_m.update(reg), ratio_m.update(reg / loss_val if loss_val else 0.0)
                    extra = dict(reg_loss=reg_m.avg, total_aux_ratio=ratio_m.avg)
                if trainer.amp16:
                    extra["skipped_steps"] = float(trainer.skipped_steps())     # the loss-scale safety net (GradScaler's skipped steps)
                write_train_logs(epoch + 1, trainer.global_step, loss_m.avg, mrr_m.avg, rec_m.avg, trainer.lr(), filename=log_file,
                                 cutoff=topk, **extra)
                for m in (loss_m, mrr_m, rec_m, reg_m, ratio_m):
                    m.reset()
            if main_rank and trainer.global_step % args.evaluate_steps == 0:
                ckpt = trainer.state_dict()
                ckpt["epoch"] = epoch + 1
                torch.save(ckpt, os.path.join(args.model_save_dir, f"checkpoint_{trainer.global_step}.pth.tar"))
    if args.distributed:
        dist.barrier()
    return trainer


def main(argv=None):
    args = set_env(get_args(argv))
    if args.rank == 0:
        os.makedirs(args.run_folder, exist_ok=True)
        import yaml
        with open(os.path.join(args.run_folder, "args.yaml"), "w") as fh:
            yaml.dump({k: (str(v) if isinstance(v, torch.device) else v) for k, v in vars(args).items()}, fh)
    owns_group = args.local_rank != -1 and dist.is_initialized()
    try:
        train(args)
    finally:
        if owns_group:                      # the group set_env() created (torchrun / --local_rank): shut RCCL down in order
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
