"""Training step of the multistep-curriculum trainers (reference ``trainer/multistep-curriculum/nway_listwise_{1,2,3}.py``,
step loop ``nway_listwise_1.py:328-367``), MI355X-native and data-parallel over RCCL.

One process per GPU (``torch.distributed`` backend "nccl" == RCCL over xGMI).  Per step, on each rank:

    forward (query tower, passage tower) -> scoring -> loss value + dlogits (one kernel) -> hand-written backward
    -> per-layer gradient buckets all-reduced on a side stream while the backward of earlier layers still runs
    -> clip_grad_norm_(max_grad_norm) as one norm pass -> legacy-HF AdamW + bf16 shadows in one pass -> linear schedule

Differences from the reference, on purpose (SURVEY.md sections 7, 8a14):
  * bf16 compute with fp32 master weights instead of fp16 autocast + GradScaler (no loss scaling is needed);
  * no per-step D2H sync: loss / train-MRR are read back only every ``logging_steps``;
  * the three stage scripts differ only in defaults, so there is one trainer with a ``--loss`` selector
    (default ``lambda_mrr`` as in the reference).
"""
from __future__ import annotations

import math

import torch
import torch.distributed as dist

from .. import hip_ops as ops
from ..models.nway_dual_encoder import NwayDualEncoder, score_mode

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


class NwayTrainer:
    def __init__(self, model: NwayDualEncoder, *, loss: str = "lambda_mrr", T: float = 1.0, learning_rate: float = 7e-6,
                 weight_decay: float = 0.01, adam_epsilon: float = 1e-8, max_grad_norm: float = 1.0, warmup_steps: int = 4000,
                 total_steps: int = 100000, betas=(0.9, 0.999), bucket_layers: int = 1):
        if loss not in LOSS_KINDS:
            raise ValueError(f"loss must be one of {LOSS_KINDS}")
        self.model = model
        self.loss_kind, self.T = loss, T
        self.lr0, self.wd, self.eps, self.max_grad_norm = learning_rate, weight_decay, adam_epsilon, max_grad_norm
        self.warmup_steps, self.total_steps, self.betas = warmup_steps, total_steps, betas
        self.global_step = 0
        self.distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.world = dist.get_world_size() if self.distributed else 1
        self.flat_p, self.flat_g = model.fuse_flat()
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
                    flags[(toff + off) // 64:(toff + off + numel + 63) // 64] = 1
            for li, (a, b) in enumerate(tower.layout.layer_range):
                self.buckets.append((ti, li, toff + a, toff + b))
            a, b = tower.layout.embed_range
            self.buckets.append((ti, -1, toff + a, toff + b))
        self.decay_flags = flags.to(dev)
        self.clip = torch.zeros(3, dtype=torch.float32, device=dev)
        self.norm_partial = torch.empty(ops.sqnorm_blocks(), dtype=torch.float32, device=dev)
        self.comm_stream = torch.cuda.Stream(device=dev) if self.distributed else None
        self.q_stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self._pending = []
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
            a, b = index[(ti, layer)]
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                dist.all_reduce(self.flat_g[a:b], op=dist.ReduceOp.SUM)
        return hook

    def forward_backward(self, batch):
        """Runs forward + backward (+ overlapped gradient all-reduce).  Returns (loss_out[2] device tensor, logits)."""
        model = self.model
        qe, pe = model.query_encoder, model.passage_encoder
        q, nw = batch["query"], batch["nway_passages"]
        bz, nway, L = nw["input_ids"].shape
        self.flat_g.zero_()
        main = torch.cuda.current_stream()
        # The query tower is ~1 % of the FLOPs but dozens of small, latency-bound launches: it runs on its own stream
        # next to the passage tower (forward and backward) instead of in front of it.
        side = self.q_stream if not model.share_weights else main
        if side is not main:
            side.wait_stream(main)
        with torch.cuda.stream(side):
            q_cls, q_tape = qe.encode(q["input_ids"], q.get("attention_mask"), train=True, save=True)
        p_cls, p_tape = pe.encode(nw["input_ids"].reshape(bz * nway, L), nw["attention_mask"].reshape(bz * nway, L),
                                  train=True, save=True)
        if side is not main:
            main.wait_stream(side)
            q_cls.record_stream(main)
        mode = score_mode(model.in_batch_loss, model.all_in_batch_neg)
        Np = nway if mode == 0 else (bz * nway if mode == 1 else 2 * nway)
        logits = torch.empty(bz, Np, dtype=torch.float32, device=q_cls.device)
        ops.score_fwd(q_cls, p_cls, logits, bz, nway, mode)
        labels = batch["labels"].to(device=logits.device, dtype=torch.float32)
        if mode != 0:   # in-batch negatives get the -0.5 label (reference nway_listwise_1.py:341-344)
            labels = torch.cat([labels, torch.full((bz, Np - nway), -0.5, dtype=torch.float32, device=logits.device)], dim=-1)
        loss_out, dlogits = ops.loss_fwd_bwd(self.loss_kind, logits, labels.contiguous(), T=self.T)
        if self.world > 1:
            dlogits.mul_(1.0 / self.world)      # gradient mean over ranks == DDP's all-reduce / world_size
        dq, dp = torch.empty_like(q_cls), torch.empty_like(p_cls)
        ops.score_bwd(dlogits, q_cls, p_cls, dq, dp, bz, nway, mode)
        if model.share_weights:
            # one tower, two tapes: gradients accumulate; all-reduce once everything is in
            pe.backward_from_cls(p_tape, dp)
            qe.backward_from_cls(q_tape, dq)
            if self.distributed:
                dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM)
        else:
            side.wait_stream(main)
            dq.record_stream(side)
            with torch.cuda.stream(side):
                qe.backward_from_cls(q_tape, dq, after_layer=self._bucket_hook(0))
            pe.backward_from_cls(p_tape, dp, after_layer=self._bucket_hook(1))
            main.wait_stream(side)
            if self.distributed:
                main.wait_stream(self.comm_stream)
        return loss_out, logits

    def optimizer_step(self):
        """clip_grad_norm_ + AdamW + scheduler.step of reference nway_listwise_1.py:355-367 (bf16: no GradScaler)."""
        self.global_step += 1
        lr = self.lr(self.global_step - 1)          # the lr in effect during this step (scheduler steps afterwards)
        ops.grad_clip_coef(self.flat_g, self.max_grad_norm, self.norm_partial, self.clip)
        towers = self.model.towers()
        # one AdamW launch over the joint buffer; it also writes the bf16 shadow of every tower
        shadow = self._joint_shadow()
        ops.adamw_step(self.flat_p, self.flat_g, self.m, self.v, self.decay_flags, shadow, lr=lr, beta1=self.betas[0],
                       beta2=self.betas[1], eps=self.eps, weight_decay=self.wd, step=self.global_step, clip=self.clip)
        for t in towers:
            t.refresh_shadows(need_transposed=True, cast=False)
        return lr

    def _joint_shadow(self):
        if getattr(self, "_shadow", None) is None:
            towers = self.model.towers()
            self._shadow = torch.empty(self.flat_p.numel(), dtype=torch.bfloat16, device=self.flat_p.device)
            for t, off in zip(towers, self.model._tower_offsets):
                t.flat_h = self._shadow[off:off + t.layout.total]
                t.refresh_shadows(need_transposed=True)
        return self._shadow

    def train_step(self, batch):
        """One full step; returns the device tensor {loss, pair count} (no host sync)."""
        self._joint_shadow()
        loss_out, logits = self.forward_backward(batch)
        self.optimizer_step()
        self.last_logits = logits
        return loss_out

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

    def state_dict(self):
        """Checkpoint payload in the reference's shape (nway_listwise_1.py:418-426): DDP-style ``module.`` prefixed keys."""
        return {"global_step": self.global_step,
                "state_dict": {"module." + k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()},
                "optimizer": {"m": self.m.cpu(), "v": self.v.cpu(), "step": self.global_step},
                "scheduler": {"last_epoch": self.global_step, "warmup_steps": self.warmup_steps, "total_steps": self.total_steps}}

    def load_state_dict(self, ckpt):
        sd = {k[7:] if k.startswith("module.") else k: v for k, v in ckpt["state_dict"].items()}
        self.model.load_state_dict(sd)
        if "optimizer" in ckpt and isinstance(ckpt["optimizer"], dict) and "m" in ckpt["optimizer"]:
            self.m.copy_(ckpt["optimizer"]["m"])
            self.v.copy_(ckpt["optimizer"]["v"])
        self.global_step = int(ckpt.get("global_step", 0))
        for t in self.model.towers():
            t.refresh_shadows(need_transposed=True)
