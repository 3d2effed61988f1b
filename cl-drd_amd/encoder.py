"""The BERT / DistilBERT encoder tower on MI355X: flat parameter store + explicit forward / backward over the
C-ABI kernels (include/cldrd_hip.h).

Stands in for ``transformers.AutoModel.from_pretrained(...)`` as used by the reference
(``models/nway_dual_encoder.py:14-19,52,56,64``): same parameter names (state-dict compatible with HF
DistilBertModel / BertModel checkpoints), same maths (post-LN transformer, erf-GELU, additive key mask,
LayerNorm eps 1e-12, dropout 0.1), CLS pooling folded into the last LayerNorm.

MI355X-first layout decisions
  * all parameters of a tower live in ONE fp32 buffer (``flat_p``; gradients in ``flat_g`` with the same layout),
    each ``nn.Parameter`` is a view -> the optimizer / gradient-norm / RCCL all-reduce run over flat ranges
    (one bucket per transformer layer, reverse order) instead of 100 small tensors;
  * q/k/v weights are adjacent, so the fused [3d, d] QKV projection is a view, not a copy;
  * the MFMA GEMMs read bf16 shadows (``flat_h``) plus K-contiguous transposed shadows of the four big matrices
    of each layer (for the data-gradient GEMMs), refreshed by the fused optimizer kernel;
  * activations are bf16, LayerNorm statistics / softmax log-sum-exp / CLS output fp32; the backward is written
    by hand (no autograd graph inside the tower), dropout masks are regenerated from a counter-based hash.
"""
from __future__ import annotations

import json
import os
from dataclasses import asdict, dataclass

import torch
import torch.nn as nn

from . import hip_ops as ops

_KNOWN = {
    "distilbert-base-uncased": dict(arch="distilbert", vocab_size=30522, dim=768, n_heads=12, hidden_dim=3072, n_layers=6,
                                    max_position_embeddings=512),
    "sebastian-hofstaetter/distilbert-dot-tas_b-b256-msmarco": dict(arch="distilbert", vocab_size=30522, dim=768, n_heads=12,
                                                                    hidden_dim=3072, n_layers=6, max_position_embeddings=512),
    "bert-base-uncased": dict(arch="bert", vocab_size=30522, dim=768, n_heads=12, hidden_dim=3072, n_layers=12,
                              max_position_embeddings=512),
}


@dataclass
class EncoderConfig:
    arch: str = "distilbert"            # "distilbert" | "bert"
    vocab_size: int = 30522
    dim: int = 768
    n_heads: int = 12
    hidden_dim: int = 3072
    n_layers: int = 6
    max_position_embeddings: int = 512
    type_vocab_size: int = 2
    eps: float = 1e-12
    dropout: float = 0.1                # hidden / embedding dropout
    attention_dropout: float = 0.1
    initializer_range: float = 0.02

    def validate(self):
        if self.arch not in ("distilbert", "bert"):
            raise ValueError(f"unsupported arch {self.arch}")
        if self.dim != self.n_heads * 64:
            raise ValueError("the attention kernel needs head dim 64 (dim == 64 * n_heads)")
        if self.dim % 128 or self.hidden_dim % 128 or self.dim > 1024:
            raise ValueError("dim and hidden_dim must be multiples of 128, dim <= 1024")

    @classmethod
    def from_hf_dict(cls, c: dict) -> "EncoderConfig":
        mt = c.get("model_type", "distilbert")
        if mt == "distilbert":
            return cls(arch="distilbert", vocab_size=c["vocab_size"], dim=c["dim"], n_heads=c["n_heads"],
                       hidden_dim=c["hidden_dim"], n_layers=c["n_layers"],
                       max_position_embeddings=c["max_position_embeddings"], dropout=c.get("dropout", 0.1),
                       attention_dropout=c.get("attention_dropout", 0.1), initializer_range=c.get("initializer_range", 0.02))
        if mt == "bert":
            return cls(arch="bert", vocab_size=c["vocab_size"], dim=c["hidden_size"], n_heads=c["num_attention_heads"],
                       hidden_dim=c["intermediate_size"], n_layers=c["num_hidden_layers"],
                       max_position_embeddings=c["max_position_embeddings"], type_vocab_size=c.get("type_vocab_size", 2),
                       eps=c.get("layer_norm_eps", 1e-12), dropout=c.get("hidden_dropout_prob", 0.1),
                       attention_dropout=c.get("attention_probs_dropout_prob", 0.1),
                       initializer_range=c.get("initializer_range", 0.02))
        raise ValueError(f"unsupported model_type {mt}")

    def to_hf_dict(self) -> dict:
        if self.arch == "distilbert":
            return dict(model_type="distilbert", vocab_size=self.vocab_size, dim=self.dim, n_heads=self.n_heads,
                        hidden_dim=self.hidden_dim, n_layers=self.n_layers,
                        max_position_embeddings=self.max_position_embeddings, dropout=self.dropout,
                        attention_dropout=self.attention_dropout, activation="gelu", initializer_range=self.initializer_range)
        return dict(model_type="bert", vocab_size=self.vocab_size, hidden_size=self.dim, num_attention_heads=self.n_heads,
                    intermediate_size=self.hidden_dim, num_hidden_layers=self.n_layers,
                    max_position_embeddings=self.max_position_embeddings, type_vocab_size=self.type_vocab_size,
                    layer_norm_eps=self.eps, hidden_dropout_prob=self.dropout,
                    attention_probs_dropout_prob=self.attention_dropout, hidden_act="gelu",
                    initializer_range=self.initializer_range)


def tiny_config() -> EncoderConfig:
    """2-layer, d = 128 DistilBERT-shaped preset (``--synthetic_model tiny`` of the trainer CLI, smoke test)."""
    return EncoderConfig(arch="distilbert", vocab_size=512, dim=128, n_heads=2, hidden_dim=256, n_layers=2,
                         max_position_embeddings=64, dropout=0.0, attention_dropout=0.0)


def layer_param_names(cfg: EncoderConfig, i: int):
    """HF names of one layer in flat-buffer order: q, k, v (weights), q, k, v (biases), out, ln1, ffn1, ffn2, ln2."""
    if cfg.arch == "distilbert":
        p = f"transformer.layer.{i}."
        q, k, v, o = p + "attention.q_lin", p + "attention.k_lin", p + "attention.v_lin", p + "attention.out_lin"
        ln1, f1, f2, ln2 = p + "sa_layer_norm", p + "ffn.lin1", p + "ffn.lin2", p + "output_layer_norm"
    else:
        p = f"encoder.layer.{i}."
        q, k, v = p + "attention.self.query", p + "attention.self.key", p + "attention.self.value"
        o, ln1 = p + "attention.output.dense", p + "attention.output.LayerNorm"
        f1, f2, ln2 = p + "intermediate.dense", p + "output.dense", p + "output.LayerNorm"
    return dict(q=q, k=k, v=v, o=o, ln1=ln1, f1=f1, f2=f2, ln2=ln2)


def param_table(cfg: EncoderConfig):
    """[(hf_name, shape)] in flat-buffer order."""
    d, f = cfg.dim, cfg.hidden_dim
    t = [("embeddings.word_embeddings.weight", (cfg.vocab_size, d)),
         ("embeddings.position_embeddings.weight", (cfg.max_position_embeddings, d))]
    if cfg.arch == "bert":
        t.append(("embeddings.token_type_embeddings.weight", (cfg.type_vocab_size, d)))
    t += [("embeddings.LayerNorm.weight", (d,)), ("embeddings.LayerNorm.bias", (d,))]
    for i in range(cfg.n_layers):
        n = layer_param_names(cfg, i)
        t += [(n["q"] + ".weight", (d, d)), (n["k"] + ".weight", (d, d)), (n["v"] + ".weight", (d, d)),
              (n["q"] + ".bias", (d,)), (n["k"] + ".bias", (d,)), (n["v"] + ".bias", (d,)),
              (n["o"] + ".weight", (d, d)), (n["o"] + ".bias", (d,)),
              (n["ln1"] + ".weight", (d,)), (n["ln1"] + ".bias", (d,)),
              (n["f1"] + ".weight", (f, d)), (n["f1"] + ".bias", (f,)),
              (n["f2"] + ".weight", (d, f)), (n["f2"] + ".bias", (d,)),
              (n["ln2"] + ".weight", (d,)), (n["ln2"] + ".bias", (d,))]
    return t


def hf_parameter_order(cfg: EncoderConfig, with_pooler: bool = True):
    """Names in the order of HF ``DistilBertModel`` / ``BertModel`` ``.named_parameters()`` (weight, bias per Linear; BERT ends with
    the pooler, which this package does not hold: the reference's optimizer counts those two in its parameter indices,
    trainer/multistep-curriculum/nway_listwise_1.py:259-263).  Checked against the installed transformers in tests/test_interop.py."""
    t = ["embeddings.word_embeddings.weight", "embeddings.position_embeddings.weight"]
    if cfg.arch == "bert":
        t.append("embeddings.token_type_embeddings.weight")
    t += ["embeddings.LayerNorm.weight", "embeddings.LayerNorm.bias"]
    for i in range(cfg.n_layers):
        n = layer_param_names(cfg, i)
        for key in ("q", "k", "v", "o", "ln1", "f1", "f2", "ln2"):
            t += [n[key] + ".weight", n[key] + ".bias"]
    if cfg.arch == "bert" and with_pooler:
        t += ["pooler.dense.weight", "pooler.dense.bias"]
    return t


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


def _env_flag(name, default):
    """A/B switch, read from the environment at every use: tests (and tools) flip them inside one process."""
    return os.environ.get(name, default)


class FlatLayout:
    """Offsets of every parameter inside the flat buffer (each start aligned to 64 elements)."""

    def __init__(self, cfg: EncoderConfig):
        self.cfg = cfg
        self.entries = {}          # name -> (offset, shape)
        self.order = []
        off = 0
        self.layer_range = []      # (start, end) of each transformer layer
        self.embed_range = None
        first_layer_name = layer_param_names(cfg, 0)["q"] + ".weight" if cfg.n_layers else None
        cur_layer_start = None
        for name, shape in param_table(cfg):
            if name == first_layer_name:
                self.embed_range = (0, off)
            for i in range(cfg.n_layers):
                if name == layer_param_names(cfg, i)["q"] + ".weight":
                    if cur_layer_start is not None:
                        self.layer_range.append((cur_layer_start, off))
                    cur_layer_start = off
            self.entries[name] = (off, tuple(shape))
            self.order.append(name)
            off += (_numel(shape) + 63) // 64 * 64
        if cur_layer_start is not None:
            self.layer_range.append((cur_layer_start, off))
        if self.embed_range is None:
            self.embed_range = (0, off)
        self.total = off
        # transposed bf16 shadows: per layer WqkvT [d,3d], WoT [d,d], W1T [d,f], W2T [f,d]
        d, f = cfg.dim, cfg.hidden_dim
        self.t_entries = []        # per layer dict name -> (offset, (rows, cols) of the transposed matrix)
        toff = 0
        for i in range(cfg.n_layers):
            e = {}
            for key, shp in (("qkv", (d, 3 * d)), ("o", (d, d)), ("f1", (d, f)), ("f2", (f, d))):
                e[key] = (toff, shp)
                toff += shp[0] * shp[1]
            self.t_entries.append(e)
        self.t_total = toff


def _attach(root: nn.Module, dotted: str, param: nn.Parameter):
    mod = root
    parts = dotted.split(".")
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, nn.Module())
        mod = mod._modules[p]
    mod.register_parameter(parts[-1], param)


class _Tape:
    """Activations saved by one training forward of one tower."""
    __slots__ = ("M", "L", "T", "ids", "mask", "seed", "mean0", "rstd0", "layers", "p_embed", "pack", "device_seed")


class _Pack:
    """Row maps of a packed batch (csrc/pack.hip): sequence m owns rows cu[m] .. cu[m + 1] of the packed matrices."""
    __slots__ = ("Tp", "cu", "tok_idx", "pos", "cls_idx", "groups")

    @classmethod
    def build(cls, lengths, L, dev):
        import numpy as np
        lens = np.asarray(lengths, dtype=np.int64).reshape(-1)
        cu = np.zeros(lens.shape[0] + 1, dtype=np.int64)
        np.cumsum(lens, out=cu[1:])
        seq = np.repeat(np.arange(lens.shape[0], dtype=np.int64), lens)
        pos = np.arange(int(cu[-1]), dtype=np.int64) - cu[seq]
        pk = cls()
        pk.Tp = int(cu[-1])
        pk.cu = torch.from_numpy(cu.astype(np.int32)).to(dev, non_blocking=True)
        pk.tok_idx = torch.from_numpy((seq * L + pos).astype(np.int32)).to(dev, non_blocking=True)
        pk.pos = torch.from_numpy(pos.astype(np.int32)).to(dev, non_blocking=True)
        pk.cls_idx = pk.cu[:-1]
        # attention launches of the batch: [(sequence list or None = all, tile height)].  At L > 128 the sequences of at most 128 tokens - most
        # of an MS MARCO batch - go through the L <= 128 kernels (persistent, double-buffered; 4 x 4 blocks at most), only the long ones through
        # the streaming forward / one-item backward of L <= 256 (cldrd_attention_*_varlen_list)
        pk.groups = [(None, 0)]
        if L > 128:
            short = np.nonzero(lens <= 128)[0]
            if short.shape[0] == lens.shape[0]:
                pk.groups = [(torch.from_numpy(short.astype(np.int32)).to(dev, non_blocking=True), 128)]
            elif short.shape[0] > 0:
                long_ = np.nonzero(lens > 128)[0]
                pk.groups = [(torch.from_numpy(short.astype(np.int32)).to(dev, non_blocking=True), 128),
                             (torch.from_numpy(long_.astype(np.int32)).to(dev, non_blocking=True), L)]
        return pk


def _drain(gen):
    """run a generator to its end; its return value"""
    try:
        while True:
            next(gen)
    except StopIteration as stop:
        return stop.value


class Stepper:
    """A forward or backward of one tower that is enqueued PIECE BY PIECE: ``step(n)`` enqueues the next n groups of launches (a group = a few
    dependent kernels: a projection + attention, an out-projection + LayerNorm, an FFN GEMM, ...), ``finish()`` the rest and returns the
    result.  The trainer uses it to enqueue the query tower's ~130 small launches in slices, each slice released by an event of the passage
    tower's stream at the start of one of ITS HBM-bound kernels (LayerNorm, attention): next to those the small GEMMs cost nothing, next
    to a large GEMM - whose grid is sized to whole rounds of 256 CUs - each of them delays a round (training step).
    ``contexts``: factories of the thread-local launch contexts (hip_ops.seed_base / loss_scale) the launches need; every step re-enters them,
    so two towers' steppers can be advanced alternately from one thread."""

    def __init__(self, gen, contexts=()):
        self._gen, self._contexts, self.done, self.result = gen, tuple(contexts), False, None

    def step(self, n: int = 1) -> bool:
        if self.done:
            return True
        from contextlib import ExitStack
        with ExitStack() as stack:
            for make in self._contexts:
                stack.enter_context(make())
            try:
                for _ in range(n):
                    next(self._gen)
            except StopIteration as stop:
                self.done, self.result = True, stop.value
        return self.done

    def finish(self):
        while not self.step(1 << 20):
            pass
        return self.result


class HipEncoder(nn.Module):
    """One encoder tower.  ``encode(ids, mask)`` -> CLS embeddings fp32 [M, d]."""

    def __init__(self, cfg: EncoderConfig, seed: int | None = None):
        super().__init__()
        cfg.validate()
        self.cfg = cfg
        self.layout = FlatLayout(cfg)
        self.flat_p = torch.zeros(self.layout.total, dtype=torch.float32)
        self.flat_g = None
        self.flat_h = None          # bf16 shadow (same layout)
        self.flat_h16 = None        # fp16 shadow (same layout): only towers that run the high-precision forward keep one
        self.hp_forward = False     # set by NwayDualEncoder on the query tower (see encode)
        self.flat_t = None          # transposed bf16 shadows
        self._t_desc = None
        self._shadow_version = -1
        self._names = list(self.layout.order)
        for name in self._names:
            off, shape = self.layout.entries[name]
            _attach(self, name, nn.Parameter(self.flat_p[off:off + _numel(shape)].view(shape)))
        self.reset_parameters(seed)
        self.step_seed = 0
        # last layer: compute only the CLS row after the K/V projection (a test hook: `cls_only_last = False` computes the full layer)
        self.cls_only_last = True
        # ---- arithmetic mode: ONE switch, CLDRD_AMP, read when the tower is built ---------------------------------------------------
        #   "fp16" (default): the reference's own mode (fp16 autocast + loss scale, nway_listwise_1.py:129,334-359).  A TRAINING pass runs
        #       every MFMA of forward and backward on fp16 operands with an fp16 tape and an fp16 gradient stream that carry a power-of-two
        #       loss scale (hip_ops.loss_scale); an EVALUATION pass reads fp16 operands in the FFN / out-projection GEMMs and in the whole
        #       query tower, bf16 in QKV / attention (fp16 QKV operands on towers deeper than 6 layers).
        #   "bf16": EVERY MFMA operand of every pass is bf16 - forward, tape, backward, both towers (BASELINE.json words cfg2-4 as bf16) -
        #       no loss scale (bf16 has fp32's range).  8-bit significands: logit drift at the level of the reference's own bf16 autocast,
        #       gradient cosines 0.99+ (tests/test_gpu_model.py states the bars of each mode).
        # Both modes keep what the reference's autocast keeps in fp32: master weights, the residual stream (pre-LN sums, LayerNorm statistics),
        # softmax, the CLS output; the gradient of the residual path is fp32 in bf16 mode (fp16 + loss scale in fp16 mode) and every
        # parameter gradient is fp32.  The operand-format sub-switches of rounds 3-5 (CLDRD_FFN_FP16, _OUT_FP16, _QKV_FP16, _QUERY_FP16,
        # _RESIDUAL, _GRAD_STREAM, _LN_ON_THE_FLY) are gone: the hybrid "bf16 tape, fp16 forward" mode they spanned is not a supported mode.
        mode = os.environ.get("CLDRD_AMP", "fp16")
        if mode not in ("fp16", "bf16"):
            raise ValueError(f"CLDRD_AMP must be fp16 or bf16, got {mode!r}")
        self.amp_mode = mode
        self._amp_fp16 = mode == "fp16"
        self.stream32 = True                  # fp32 residual stream (pre-LN sums) in both modes
        self.grad_stream32 = True             # fp32 arithmetic on the residual path of the backward in both modes
        self.ffn_fp16 = self._amp_fp16        # evaluation pass of the fp16 mode: FFN / out-projection (and deep towers' QKV) GEMMs on fp16 operands
        self._grad_stream16 = True            # fp16 mode: the gradient stream is stored in fp16 between kernels (test hook: False keeps it fp32)
        self.ln_defer = True                  # LayerNorm-parameter gradients reduced by one grouped launch per flush (test hook: False = immediately)

    # ------------------------------------------------------------------ parameters
    def named_flat(self):
        d = dict(self.named_parameters())
        return [(n, d[n]) for n in self._names]

    def reset_parameters(self, seed: int | None = None):
        """HF init: N(0, initializer_range) for Linear / Embedding weights, LN gamma 1, biases and LN beta 0."""
        g = torch.Generator().manual_seed(0 if seed is None else seed)
        with torch.no_grad():
            for name, p in self.named_flat():
                if "LayerNorm.weight" in name or "layer_norm.weight" in name:
                    p.fill_(1.0)
                elif name.endswith(".bias"):
                    p.zero_()
                else:
                    p.copy_(torch.randn(p.shape, generator=g) * self.cfg.initializer_range)

    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        self._reflatten()
        return self

    def _rebind(self, flat: torch.Tensor):
        """Make every parameter a TRUE view of ``flat`` (fresh ``nn.Parameter`` objects created from the view): a Parameter made
        from a view shares the base tensor's version counter, so ``optimizer.step()`` / ``load_state_dict`` / any in-place
        write through a parameter bumps ``flat_p._version`` and the bf16 shadows are known to be stale.  (Rebinding with
        ``param.data = view`` would give each parameter a private counter: torch 2.10.)"""
        self.__dict__.pop("_view_cache", None)         # cached parameter views belong to the buffers being replaced
        old = dict(self.named_parameters())
        for n in self._names:
            off, shape = self.layout.entries[n]
            _attach(self, n, nn.Parameter(flat[off:off + _numel(shape)].view(shape), requires_grad=old[n].requires_grad))
        self.flat_p = flat

    def _reflatten(self):
        """Re-establish the flat aliasing after .to()/.cuda() moved the parameters one by one."""
        params = dict(self.named_parameters())
        dev = params[self._names[0]].device
        if self.flat_p.device != dev or any(
                params[n].data_ptr() != self.flat_p.data_ptr() + 4 * self.layout.entries[n][0] for n in self._names[:3]):
            flat = torch.zeros(self.layout.total, dtype=torch.float32, device=dev)
            for n in self._names:
                off, shape = self.layout.entries[n]
                flat[off:off + _numel(shape)].view(shape).copy_(params[n].data.to(torch.float32))
            self._rebind(flat)
            self.flat_g = None
            self.flat_h = self.flat_h16 = self.flat_t = self._t_desc = None
            self._shadow_version = -1

    def adopt_flat(self, flat_p: torch.Tensor, flat_g: torch.Tensor | None = None):
        """Move this tower's parameters into a slice of a model-level flat buffer (same device)."""
        assert flat_p.numel() == self.layout.total and flat_p.dtype == torch.float32
        self.__dict__.pop("_view_cache", None)
        flat_p.copy_(self.flat_p.to(flat_p.device))
        self._rebind(flat_p)
        self.flat_g = flat_g
        if flat_g is not None:
            self._bind_grads()
        self._shadow_version = -1

    def _bind_grads(self):
        params = dict(self.named_parameters())
        for n in self._names:
            off, shape = self.layout.entries[n]
            params[n].grad = self.flat_g[off:off + _numel(shape)].view(shape)

    def ensure_grads(self, check_all: bool = False):
        """``param.grad`` of every parameter is a view of ``flat_g``.  ``check_all`` (autograd bridge, i.e. the reference-style
        ``loss.backward()`` loop): ``optimizer.zero_grad()`` / ``model.zero_grad()`` default to ``set_to_none=True``, which drops
        the views; the gradients they would have held are stale then, so the flat buffer is zeroed before the views come back
        (once per step: the second tape of a shared tower finds the views in place and accumulates)."""
        if self.flat_g is None or self.flat_g.device != self.flat_p.device:
            self.flat_g = torch.zeros_like(self.flat_p)
            self._bind_grads()
            return
        if check_all:
            params = dict(self.named_parameters())
            missing = []
            for n in self._names:
                off, _ = self.layout.entries[n]
                gr = params[n].grad
                if gr is None or gr.data_ptr() != self.flat_g.data_ptr() + 4 * off:
                    missing.append(n)
            if len(missing) == len(self._names):
                self.flat_g.zero_()
            else:
                for n in missing:
                    self.g(n).zero_()
            if missing:
                self._bind_grads()
            return
        p0 = next(self.parameters())
        if p0.grad is None or p0.grad.data_ptr() != self.flat_g.data_ptr():
            self._bind_grads()

    # Views of one parameter inside the flat buffers.  They are looked up ~250 times per training step; slicing + reshaping a tensor
    # costs ~2.5 us, so the views are cached per (buffer identity, name) - a flat buffer that is replaced (`.to()`, adopt_flat,
    # a new shadow) has another data_ptr and simply misses.
    def _view(self, flat, tag, name):
        cache = self.__dict__.setdefault("_view_cache", {})
        key = (tag, name)
        hit = cache.get(key)
        ptr = flat.data_ptr()
        if hit is not None and hit[0] == ptr:
            return hit[1]
        off, shape = self.layout.entries[name]
        v = flat[off:off + _numel(shape)].view(shape)
        cache[key] = (ptr, v)
        return v

    def g(self, name):
        return self._view(self.flat_g, "g", name)

    def w(self, name):
        return self._view(self.flat_p, "p", name)

    def h(self, name):
        return self._view(self.flat_h, "h", name)

    # ------------------------------------------------------------------ bf16 shadows
    def _build_t_desc(self):
        cfg, lay = self.cfg, self.layout
        desc, prefix, tiles = [], [0], 0
        for i in range(cfg.n_layers):
            n = layer_param_names(cfg, i)
            te = lay.t_entries[i]
            for key, src_name, rows, cols in (("qkv", n["q"] + ".weight", 3 * cfg.dim, cfg.dim),
                                              ("o", n["o"] + ".weight", cfg.dim, cfg.dim),
                                              ("f1", n["f1"] + ".weight", cfg.hidden_dim, cfg.dim),
                                              ("f2", n["f2"] + ".weight", cfg.dim, cfg.hidden_dim)):
                desc += [lay.entries[src_name][0], te[key][0], rows, cols]
                tiles += ((rows + 31) // 32) * ((cols + 31) // 32)
                prefix.append(tiles)
        dev = self.flat_p.device
        self._t_desc = (torch.tensor(desc, dtype=torch.int64, device=dev), torch.tensor(prefix, dtype=torch.int32, device=dev),
                        len(desc) // 4, tiles)
        # the same batch for the bf16 -> bf16 kernel (64 x 64 tiles) when every matrix allows it
        self._t_desc64 = None
        if all(desc[k + 2] % 64 == 0 and desc[k + 3] % 64 == 0 and desc[k] % 8 == 0 and desc[k + 1] % 8 == 0 for k in range(0, len(desc), 4)):
            p64, t64 = [0], 0
            for k in range(0, len(desc), 4):
                t64 += (desc[k + 2] // 64) * (desc[k + 3] // 64)
                p64.append(t64)
            self._t_desc64 = (torch.tensor(p64, dtype=torch.int32, device=dev), t64)

    @property
    def amp16(self):
        """the all-fp16 training mode (see __init__): needs the fp32 residual stream and the fp32 gradient stream"""
        return self._amp_fp16 and self.stream32 and self.grad_stream32

    @property
    def grad_stream16(self):
        """the gradient stream is fp16 between kernels (round 5; only in the all-fp16 training mode, see __init__)"""
        return self.amp16 and self._grad_stream16

    @property
    def needs_h16(self):
        return self.hp_forward or self.ffn_fp16 or self.amp16

    def h16_buffer(self):
        """The fp16 weight shadow of a tower whose forward reads fp16 weights (the high-precision pass of the query tower; the FFN GEMMs
        of every tower by default), allocated on first use; else None."""
        if not self.needs_h16:
            return None
        if self.flat_h16 is None or self.flat_h16.device != self.flat_p.device:
            self.flat_h16 = torch.empty(self.layout.total, dtype=torch.float16, device=self.flat_p.device)
        return self.flat_h16

    def refresh_shadows(self, need_transposed: bool = True, cast: bool = True, cast16: bool | None = None, h_stale: bool = False):
        """bf16 copies of the weights for the MFMA GEMMs (+ transposed copies for the data-gradient GEMMs).  ``cast`` / ``cast16`` False:
        the optimizer step has already written the bf16 / fp16 shadow (``cast16`` None = cast the fp16 shadow whenever the tower has one).
        ``h_stale`` (with ``cast=False``): the optimizer step did NOT write the bf16 shadow - an all-fp16 training step reads none of it
        (172 MB less written per step); the next pass that does (an evaluation forward) casts it first (`_shadows_ok`)."""
        if not self.flat_p.is_cuda:
            raise RuntimeError("HipEncoder runs on the GPU only: move the model with .cuda() first (no CPU path)")
        if self.flat_h is None:
            self.flat_h = torch.empty(self.layout.total, dtype=torch.bfloat16, device=self.flat_p.device)
        if cast:
            ops.cast_bf16(self.flat_p, self.flat_h)
            self._h_stale = False
        elif h_stale:
            self._h_stale = True
        if self.needs_h16 and (cast16 is None or cast16):
            ops.cast_f16(self.flat_p, self.h16_buffer())
        if need_transposed and self.cfg.n_layers:
            self._transpose_shadows()
        else:
            self._t_fresh = False
        self._shadow_version = self.flat_p._version

    def refresh_transposed(self):
        """Only the transposed bf16 copies (the B operands of the data-gradient GEMMs) from the current bf16 shadow: the trainer makes
        them at the START of the next step on its second stream - nothing reads them before the backward - instead of after AdamW
        on the main one (``refresh_shadows(need_transposed=False)`` there)."""
        if not self.cfg.n_layers:
            self._t_fresh = True
            return
        self._transpose_shadows()

    def _transpose_shadows(self):
        """flat_t: K-contiguous copies of the four matrices of every layer for the data-gradient GEMMs, in the backward's operand format -
        bf16 from the bf16 shadow, or (amp16) fp16 from the fp16 shadow.  Both shadows are current when this runs."""
        t16 = torch.float16 if self.amp16 else torch.bfloat16
        if self.flat_t is None or self.flat_t.dtype != t16:
            self.flat_t = torch.empty(self.layout.t_total, dtype=t16, device=self.flat_p.device)
        if self._t_desc is None:
            self._build_t_desc()
        desc, prefix, nd, tiles = self._t_desc
        if self._t_desc64 is not None:
            ops.transpose_bf16_batched(self.h16_buffer() if self.amp16 else self.flat_h, self.flat_t, desc, self._t_desc64[0], nd, self._t_desc64[1])
        elif self.amp16:
            raise RuntimeError("CLDRD_AMP=fp16 needs layer matrices whose sides are multiples of 64 (the 16-bit transpose kernel)")
        else:
            ops.transpose_cast_batched(self.flat_p, self.flat_t, desc, prefix, nd, tiles)
        self._t_fresh = True

    def _shadows_ok(self, need_t, need_h=True):
        return (self.flat_h is not None and self._shadow_version == self.flat_p._version and (not self.needs_h16 or self.flat_h16 is not None) and
                (not need_h or not getattr(self, "_h_stale", False)) and
                (not need_t or (self.flat_t is not None and getattr(self, "_t_fresh", False))))

    def ht(self, layer, key):
        off, shp = self.layout.t_entries[layer][key]
        return self.flat_t[off:off + shp[0] * shp[1]].view(shp)

    # ------------------------------------------------------------------ forward
    def _layer_weights(self, i, fp16=False):
        sh0 = self.flat_h16 if fp16 else self.flat_h
        ck = ("W", i, bool(fp16))
        cache = self.__dict__.setdefault("_view_cache", {})
        hit = cache.get(ck)
        if hit is not None and hit[0] == (sh0.data_ptr(), self.flat_p.data_ptr()):
            return hit[1]
        W = self._layer_weights_build(i, fp16)
        cache[ck] = ((sh0.data_ptr(), self.flat_p.data_ptr()), W)
        return W

    def _layer_weights_build(self, i, fp16=False):
        cfg, n = self.cfg, layer_param_names(self.cfg, i)
        d = cfg.dim
        oq = self.layout.entries[n["q"] + ".weight"][0]
        ob = self.layout.entries[n["q"] + ".bias"][0]
        sh = self.flat_h16 if fp16 else self.flat_h

        def h(name):
            off, shape = self.layout.entries[name]
            return sh[off:off + _numel(shape)].view(shape)
        return dict(Wqkv=sh[oq:oq + 3 * d * d].view(3 * d, d), bqkv=self.flat_p[ob:ob + 3 * d],
                    Wo=h(n["o"] + ".weight"), bo=self.w(n["o"] + ".bias"),
                    g1=self.w(n["ln1"] + ".weight"), b1=self.w(n["ln1"] + ".bias"),
                    W1=h(n["f1"] + ".weight"), bf1=self.w(n["f1"] + ".bias"),
                    W2=h(n["f2"] + ".weight"), bf2=self.w(n["f2"] + ".bias"),
                    g2=self.w(n["ln2"] + ".weight"), b2=self.w(n["ln2"] + ".bias"))

    def _layer_grads(self, i):
        ck = ("G", i)
        cache = self.__dict__.setdefault("_view_cache", {})
        hit = cache.get(ck)
        if hit is not None and hit[0] == self.flat_g.data_ptr():
            return hit[1]
        G = self._layer_grads_build(i)
        cache[ck] = (self.flat_g.data_ptr(), G)
        return G

    def _layer_grads_build(self, i):
        cfg, n = self.cfg, layer_param_names(self.cfg, i)
        d = cfg.dim
        oq = self.layout.entries[n["q"] + ".weight"][0]
        ob = self.layout.entries[n["q"] + ".bias"][0]
        return dict(Wqkv=self.flat_g[oq:oq + 3 * d * d].view(3 * d, d), bqkv=self.flat_g[ob:ob + 3 * d],
                    Wo=self.g(n["o"] + ".weight"), bo=self.g(n["o"] + ".bias"),
                    g1=self.g(n["ln1"] + ".weight"), b1=self.g(n["ln1"] + ".bias"),
                    W1=self.g(n["f1"] + ".weight"), bf1=self.g(n["f1"] + ".bias"),
                    W2=self.g(n["f2"] + ".weight"), bf2=self.g(n["f2"] + ".bias"),
                    g2=self.g(n["ln2"] + ".weight"), b2=self.g(n["ln2"] + ".bias"))

    @staticmethod
    def _buf(rows, cols, dev, dtype=torch.bfloat16):
        # rows rounded up to 64: allocation granularity only (no kernel reads past `rows`; the weight-gradient kernel takes the
        # missing token rows of its last K tile from a zero page, so nothing is zero-filled here - round 1 spent 102 fills a step)
        return torch.empty(ops.pad_rows(rows), cols, dtype=dtype, device=dev)

    def encode(self, input_ids: torch.Tensor, attention_mask: torch.Tensor | None, *, train: bool | None = None,
               save: bool = False, seed: int | None = None, fp16: bool | None = None, lengths=None, device_seed: bool = False):
        """CLS embeddings fp32 [M, d] (== HF ``model(**enc)[0][:, 0, :]``).  With ``save`` also returns the tape.

        ``fp16`` (default: ``self.hp_forward``, which NwayDualEncoder sets on the QUERY tower in the fp16 mode): the evaluation pass runs
        every GEMM and the attention on fp16 operands (11-bit significands, same MFMA rate; forward activations of a BERT encoder are
        far inside the fp16 range, and the reference itself runs fp16 autocast on its GPUs).  Why the query tower: a logit error is
        dq.p + q.dp, there are N passages per query, and with CLS vectors that share a large common component every logit of a row
        inherits the SAME dq.p term - B draws dominate max|dlogit| - while the query tower is ~1 % of the FLOPs.  Ignored in the bf16 mode.

        ``lengths`` (host-side ints, one per sequence: the number of leading 1s of its mask row): with them a padded batch is PACKED -
        the Linear / LayerNorm / weight-gradient kernels run on the real tokens only (the reference pads every sequence to the longest of
        the batch and computes on the padding: ~40 % of the rows of an MS MARCO batch); attention keeps the padded layout; the CLS output
        is the same up to the order of fp32 summation.  Without them nothing is packed (finding the row count would cost a host sync)."""
        return self.encode_steps(input_ids, attention_mask, train=train, save=save, seed=seed, fp16=fp16, lengths=lengths,
                                 device_seed=device_seed).finish()

    def encode_steps(self, input_ids, attention_mask, *, train=None, save=False, seed=None, fp16=None, lengths=None, device_seed=False,
                     window=None) -> Stepper:
        """:meth:`encode` as a :class:`Stepper` (``.finish()`` -> what encode returns).  ``window(kind)`` (optional) is called right before
        each HBM-bound launch of the pass - kind "ln" (LayerNorm / embedding kernels) or "attn" - on the stream the pass is enqueued on: the
        hook at which a trainer releases a slice of the OTHER tower's stepper."""
        fp16 = self.hp_forward if fp16 is None else fp16
        base = getattr(self, "seed_base_ptr", None) if (device_seed and seed is None) else None
        if base:
            # graph mode, NwayTrainer only (`device_seed`): the per-step part of the seed lives in device memory at `seed_base_ptr` (the
            # trainer advances step_seed and writes next_seed() there before each step); the launches carry offsets only.  Any other
            # caller (encode_autograd, an explicit seed) gets a by-value seed as before - the device word is only advanced by the trainer.
            gen = self._encode_pair(input_ids, attention_mask, train=train, save=save, seed=0, fp16=fp16, lengths=lengths, window=window)

            def mark(gen=gen):
                out = yield from gen
                if save:
                    out[1].device_seed = True           # the backward adds the same device word
                return out
            return Stepper(mark(), [lambda: ops.seed_base(base)])
        if seed is None:
            seed = self.next_seed()
        return Stepper(self._encode_pair(input_ids, attention_mask, train=train, save=save, seed=seed, fp16=fp16, lengths=lengths, window=window))

    def would_pack(self, lengths, M, L, has_mask=True, fp16=False, train=True) -> bool:
        """Whether encode() packs a batch with these host-side token counts (CLDRD_PACK=0 turns packing off; thresholds at the
        end: a training batch with less than 8 % padding keeps its graph replay, an evaluation batch packs from 3 % padding).  The trainer asks BEFORE choosing between the replayed graph and the
        eager step: only a batch that really is packed changes its row count from step to step."""
        if (lengths is None or not has_mask or (fp16 and not self.amp16) or not self.cls_only_last or self.cfg.n_layers < 1 or L <= 1
                or _env_flag("CLDRD_PACK", "1") == "0"):
            return False
        lens = [int(v) for v in (lengths.reshape(-1).tolist() if hasattr(lengths, "reshape") else lengths)]
        if len(lens) != M or min(lens) < 1 or max(lens) > L:
            # the packed kernels take every sequence to own 1 .. L rows (cldrd_attention_*_varlen, CLS row = first row): anything else - an empty
            # row of a partially filled batch, counts that do not belong to this batch - stays on the padded path, which reads the mask
            return False
        n_tok = sum(lens)
        if n_tok < 1024 <= M * L:
            # packing would move the Linear layers from the large-M GEMM kernel to the small-M one (split along K: another summation order), and
            # "packed == padded bit for bit" would stop holding (tools/model_fuzz.py, seed 1 case 0); at such sizes packing buys nothing
            return False
        # a training step that packs cannot replay its HIP graph (the row count changes every step): only with 8 % padding or more.  An
        # evaluation pass (index encode) launches eagerly anyway and packing moves no rows any more (attention reads packed rows): 3 %
        return 0 < n_tok <= int((0.92 if train else 0.97) * M * L)

    def next_seed(self) -> int:
        """Advance the step counter; the dropout seed of the step (what encode() draws when no seed is given)."""
        self.step_seed += 1
        return (self.step_seed * 0x9E3779B1) & 0x7FFFFFFFFFFF

    def _encode_pair(self, input_ids, attention_mask, *, train, save, seed, fp16, lengths=None, window=None):
        """The generator of the pass that runs (see __init__, CLDRD_AMP).  fp16 mode: a pass that keeps a tape is the all-fp16 pass; an
        evaluation pass of the query tower (``fp16``: hp_forward) is the all-fp16 pass too, any other evaluation pass reads fp16 operands in
        the FFN / out-projection GEMMs and bf16 in QKV / attention.  bf16 mode: one pass, every operand bf16 (``fp16`` is ignored)."""
        if self.amp16:
            if save:
                return self._encode_gen(input_ids, attention_mask, train=train, save=True, seed=seed, fp16=True, lengths=lengths, window=window)
            if fp16 and self.hp_forward and input_ids.dim() == 2 and input_ids.shape[1] <= 128:
                return self._encode_gen(input_ids, attention_mask, train=train, save=False, seed=seed, fp16=True, window=window)
        return self._encode_gen(input_ids, attention_mask, train=train, save=save, seed=seed, fp16=False, lengths=lengths, window=window)

    def _encode(self, input_ids, attention_mask, *, train, save, seed, fp16, lengths=None):
        return _drain(self._encode_gen(input_ids, attention_mask, train=train, save=save, seed=seed, fp16=fp16, lengths=lengths))

    def _encode_gen(self, input_ids, attention_mask, *, train, save, seed, fp16, lengths=None, window=None):
        """one forward pass as a generator: a ``yield`` behind every group of launches (:class:`Stepper`); returns cls or (cls, tape)"""
        cfg = self.cfg
        train = self.training if train is None else train
        if input_ids.dim() != 2:
            raise ValueError("input_ids must be [M, L]")
        M, L = input_ids.shape
        if L > 256 or L > cfg.max_position_embeddings:
            raise ValueError("sequence length must be <= 256 (and <= max_position_embeddings)")
        dev = self.flat_p.device
        if not self._shadows_ok(save, need_h=not (self.amp16 and save)):      # (the all-fp16 training pass reads the fp16 shadow only)
            self.refresh_shadows(need_transposed=save)
        ids = input_ids.to(device=dev, dtype=torch.int64).contiguous()
        mask = None if attention_mask is None else attention_mask.to(device=dev, dtype=torch.int64).contiguous()
        T, d, f, H = M * L, cfg.dim, cfg.hidden_dim, cfg.n_heads
        p_h = cfg.dropout if train else 0.0
        p_a = cfg.attention_dropout if train else 0.0
        dt16 = torch.float16 if fp16 else torch.bfloat16         # 16-bit activation format of this pass
        pk = None
        if lengths is not None and len(lengths) != M:
            raise ValueError("lengths: one entry per sequence")
        if self.would_pack(lengths, M, L, has_mask=mask is not None, fp16=fp16, train=save):
            pk = _Pack.build(lengths, L, dev)
            T = pk.Tp
            ids_padded, ids = ids, ops.gather_i64(ids.reshape(-1), pk.tok_idx, T)
        tape = None
        if save:
            tape = _Tape()
            tape.M, tape.L, tape.T, tape.ids, tape.mask, tape.seed, tape.layers = M, L, T, ids, mask, seed, []
            tape.p_embed, tape.pack, tape.device_seed = p_h, pk, False
        f32 = dict(dtype=torch.float32, device=dev)
        sdt = torch.float32                                     # storage type of the pre-LN sums: the fp32 residual stream
        # QKV16 (evaluation pass of the fp16 mode, towers deeper than 6 layers): the QKV projection reads fp16 operands too (bf16 q / k / v
        # out: that pass's attention kernels are bf16).  With the FFN pair in fp16, x_in / Wqkv are the next largest source of logit drift
        # (BERT-base cfg4 golden: 0.190 -> 0.116); the 6-layer configs are at 2-3.5e-3 without it and would only pay for it.
        QKV16 = self.ffn_fp16 and not fp16 and cfg.n_layers > 6
        if QKV16 and save:
            raise RuntimeError("a pass that keeps a tape runs one operand format (fp16 or bf16), never the evaluation pass's mixed one")
        x = self._buf(T, d, dev, dt16) if not QKV16 else None
        xh = self._buf(T, d, dev, torch.float16) if QKV16 else None
        x32 = self._buf(T, d, dev, torch.float32)           # fp32 copy of the layer input: the residual operand
        mean0, rstd0 = torch.empty(T, **f32), torch.empty(T, **f32)
        type0 = self.w("embeddings.token_type_embeddings.weight")[0] if cfg.arch == "bert" else None
        if window is not None:
            window("ln")
        ops.embed_ln_fwd(ids.view(-1), self.w("embeddings.word_embeddings.weight"),
                         self.w("embeddings.position_embeddings.weight"), type0, self.w("embeddings.LayerNorm.weight"),
                         self.w("embeddings.LayerNorm.bias"), xh if QKV16 else x, mean0, rstd0, T, L, cfg.eps, p_h, seed, out32=x32,
                         pos_idx=pk.pos if pk is not None else None)
        if save:
            tape.mean0, tape.rstd0 = mean0, rstd0
        yield
        cls = torch.empty(M, d, **f32)
        p_out = p_h if cfg.arch == "bert" else 0.0          # DistilBERT has no dropout after out_lin
        # fp32 residual stream: a LayerNorm's fp32 output is consumed exactly once, as the residual operand of the next out-projection /
        # FFN2 epilogue.  It is not stored: that epilogue reads the pre-LN sum (which the backward keeps anyway) and applies mean / rstd /
        # gamma / beta on the fly (`residual_ln`) - 100 MB less written per LayerNorm at cfg2, the same bytes read.
        res_ln = None                                        # LayerNorm still to be applied to x32 (None: x32 is the value itself)
        FFN16 = self.ffn_fp16 and not fp16              # fp16 mode, evaluation pass: the FFN and out-projection GEMMs read fp16 operands (the
        OUT16 = FFN16                                   # attention kernel leaves its context in fp16 for the out-projection)
        if FFN16 and save:
            raise RuntimeError("a pass that keeps a tape runs one operand format (fp16 or bf16), never the evaluation pass's mixed one")
        for i in range(cfg.n_layers):
            W = self._layer_weights(i, fp16)
            W16 = self._layer_weights(i, True) if FFN16 else None
            s_l = seed + 7919 * (i + 1)
            if i == cfg.n_layers - 1 and self.cls_only_last:
                yield from self._last_layer_cls_fwd(x, x32, W, mask, M, L, T, p_h, p_a, p_out, s_l, save, tape, cls, dt16, W16, pk, xh, out16=OUT16)
                break
            qkv = self._buf(T, 3 * d, dev, dt16)
            if QKV16:
                ops.gemm_nt(xh, W16["Wqkv"], qkv, T, bias=W["bqkv"])          # fp16 operands, bf16 q / k / v
            else:
                ops.gemm_nt(x, W["Wqkv"], qkv, T, bias=W["bqkv"])
            lse = torch.empty(M, H, L, **f32) if save else None
            dbits = ops.attention_drop_bits(M, L, H, p_a, dev) if (save and (dt16 == torch.bfloat16 or self.amp16)) else None      # dropout keep bits for the backward
            # OUT16: the out-projection on fp16 operands too - the attention kernel leaves its context in fp16 next to (training) or
            # instead of (evaluation) the bf16 tensor the backward's MFMAs read.  With the FFN pair already on fp16 operands this is the
            # rounding point that carries most of what is left of the logit drift (CPU emulation on the goldens, DESIGN.md section 2:
            # BERT-base 5.25e-3 -> 2.08e-3 of the logit scale, DistilBERT 2.98e-3 -> 2.05e-3).
            ctx16 = self._buf(T, d, dev, torch.float16) if OUT16 else None
            # packed batch: the attention kernels find a sequence's rows through cu (keys beyond its length masked, padding rows neither
            # loaded nor stored: cldrd_attention_*_varlen) - bit for bit what moving the rows to the padded [M * L, .] layout and back
            # gave until round 6 (two row moves per layer and pass, 0.56 ms of an 8.3-ms MS MARCO-shaped training step)
            cu = pk.cu if pk is not None else None
            ctx = self._buf(T, d, dev, dt16) if (save or not OUT16) else None
            yield
            if window is not None:
                window("attn")
            for sl, tile in (pk.groups if pk is not None else [(None, 0)]):
                ops.attention_fwd(qkv, mask if pk is None else None, ctx, lse, M, L, H, p_a, s_l + 1, drop_bits=dbits, ctx16=ctx16,
                                  full_family=fp16 and self.amp16, cu=cu, seq_list=sl, tile=tile)
            yield
            s1 = self._buf(T, d, dev, sdt)
            ops.gemm_nt(ctx16 if OUT16 else ctx, (W16 if OUT16 else W)["Wo"], s1, T, bias=W["bo"], residual=x32, dropout_p=p_out,
                        seed=s_l + 2, residual_ln=res_ln)
            del ctx16
            yield
            if window is not None:
                window("ln")
            mean1, rstd1 = torch.empty(T, **f32), torch.empty(T, **f32)
            if FFN16:
                # evaluation pass of the fp16 mode: LayerNorm -> fp16 (FFN1's operand); FFN1 -> h in fp16 (FFN2's operand); no tape
                x1 = hbuf = pre = None
                x1h = self._buf(T, d, dev, torch.float16)
                ops.layernorm_fwd(s1, W["g1"], W["b1"], x1h, mean1, rstd1, T, cfg.eps)
                yield
                hh = self._buf(T, f, dev, torch.float16)
                ops.gemm_nt(x1h, W16["W1"], hh, T, bias=W["bf1"], act=1)
                yield
                s2 = self._buf(T, d, dev, sdt)
                ops.gemm_nt(hh, W16["W2"], s2, T, bias=W["bf2"], residual=s1, dropout_p=p_h, seed=s_l + 3, residual_ln=(mean1, rstd1, W["g1"], W["b1"]))
                del x1h, hh
            else:
                x1 = self._buf(T, d, dev, dt16)
                ops.layernorm_fwd(s1, W["g1"], W["b1"], x1, mean1, rstd1, T, cfg.eps)
                yield
                hbuf = self._buf(T, f, dev, dt16)
                pre = self._buf(T, f, dev, dt16) if save else None
                ops.gemm_nt(x1, W["W1"], hbuf, T, bias=W["bf1"], preact=pre, act=3 if save else 1)     # the tape keeps gelu'(pre-activation)
                yield
                s2 = self._buf(T, d, dev, sdt)
                ops.gemm_nt(hbuf, W["W2"], s2, T, bias=W["bf2"], residual=s1, dropout_p=p_h, seed=s_l + 3, residual_ln=(mean1, rstd1, W["g1"], W["b1"]))
            xo = self._buf(T, d, dev, dt16) if not QKV16 else None
            xoh = self._buf(T, d, dev, torch.float16) if QKV16 else None
            last = i == cfg.n_layers - 1
            # the CLS-only last layer gathers its fp32 residual rows from a stored copy: the layer before it still writes one
            need32 = not last and i + 1 == cfg.n_layers - 1 and self.cls_only_last
            xo32 = self._buf(T, d, dev, torch.float32) if need32 else None
            mean2, rstd2 = torch.empty(T, **f32), torch.empty(T, **f32)
            yield
            if window is not None:
                window("ln")
            ops.layernorm_fwd(s2, W["g2"], W["b2"], xoh if QKV16 else xo, mean2, rstd2, T, cfg.eps, cls if last else None, L, out32=xo32)
            if save:
                tape.layers.append(dict(x_in=x, qkv=qkv, ctx=ctx, lse=lse, dbits=dbits, s1=s1, mean1=mean1, rstd1=rstd1, x1=x1, pre=pre,
                                        h=hbuf, s2=s2, mean2=mean2, rstd2=rstd2, seed=s_l, p_h=p_h, p_a=p_a, p_out=p_out))
            x, xh = xo, xoh
            x32, res_ln = (xo32, None) if need32 else (s2, (mean2, rstd2, W["g2"], W["b2"]))
            yield
        if cfg.n_layers == 0:
            src = x if x is not None else xh
            cls.copy_(src.view(ops.pad_rows(T), d)[:T].view(M, L, d)[:, 0].float())
        return (cls, tape) if save else cls

    # ------------------------------------------------------------------ last layer, CLS row only
    def _last_layer_cls_fwd(self, x, x32, W, mask, M, L, T, p_h, p_a, p_out, s_l, save, tape, cls, dt16=torch.bfloat16, W16=None, pk=None, xh=None,
                            out16=False):
        """Only ``last_hidden_state[:, 0, :]`` is consumed (reference models/nway_dual_encoder.py:52,56,64), so the last
        layer projects K and V for every token but Q, attention, out-proj, FFN and both LayerNorms for token 0 only:
        identical CLS output, ~1/6 of the layer's FLOPs."""
        cfg = self.cfg
        d, f, H = cfg.dim, cfg.hidden_dim, cfg.n_heads
        dev = (x if x is not None else xh).device
        f32 = dict(dtype=torch.float32, device=dev)
        kv = self._buf(T, 2 * d, dev, dt16)
        Q16 = xh is not None and W16 is not None              # fp16 operands for the K / V / Q projections (bf16 results), see _encode
        if Q16:
            ops.gemm_nt(xh, W16["Wqkv"][d:], kv, T, bias=W["bqkv"][d:])
        else:
            ops.gemm_nt(x, W["Wqkv"][d:], kv, T, bias=W["bqkv"][d:])
        yield
        sdt = torch.float32
        if pk is None:
            # the CLS rows are rows 0, L, 2L, ... of the layer input: STRIDED VIEWS (row pitch L * d), no copies - every consumer (the Q
            # projection's A operand, its weight gradient's X operand, the out-projection's fp32 residual) takes a row pitch.  Until round 5
            # two or three 6-us gather launches sat here, on the critical path between the towers' forward and the loss
            xc = x[:T].view(M, L * d)[:, :d] if x is not None else None
            xch = xh[:T].view(M, L * d)[:, :d] if Q16 else None
            xc32 = x32[:T].view(M, L * d)[:, :d]
        else:
            xc = self._buf(M, d, dev, dt16) if x is not None else None          # 16-bit CLS rows: the weight gradient's operand (and, without Q16, the GEMM's)
            xch = self._buf(M, d, dev, torch.float16) if Q16 else None
            xc32 = self._buf(M, d, dev, torch.float32)
            # packed batch: the CLS token of sequence m is row cu[m]; the CLS attention kernel reads the packed K / V rows through cu
            if xc is not None:
                ops.gather_rows(x, pk.cls_idx, xc, M)
            if Q16:
                ops.gather_rows(xh, pk.cls_idx, xch, M)
            ops.gather_rows(x32, pk.cls_idx, xc32, M)
        qc = self._buf(M, d, dev, dt16)
        if Q16:
            ops.gemm_nt(xch, W16["Wqkv"][:d], qc, M, bias=W["bqkv"][:d])
        else:
            ops.gemm_nt(xc, W["Wqkv"][:d], qc, M, bias=W["bqkv"][:d])
        out16 = out16 and W16 is not None                    # out-projection on fp16 operands (see _encode_gen)
        ctxc = self._buf(M, d, dev, dt16) if (save or not out16) else None
        ctxc16 = self._buf(M, d, dev, torch.float16) if out16 else None
        probs = torch.empty(M, H, L, **f32)
        ops.attention_cls_fwd(qc, kv, mask if pk is None else None, ctxc, probs, M, L, H, p_a, s_l + 1, ctx16=ctxc16,
                              cu=pk.cu if pk is not None else None)
        yield
        s1 = self._buf(M, d, dev, sdt)
        ops.gemm_nt(ctxc16 if out16 else ctxc, (W16 if out16 else W)["Wo"], s1, M, bias=W["bo"], residual=xc32, dropout_p=p_out, seed=s_l + 2)
        yield
        x1_32 = self._buf(M, d, dev, torch.float32)
        mean1, rstd1 = (torch.empty(M, **f32), torch.empty(M, **f32)) if save else (None, None)
        if W16 is not None:                 # evaluation pass of the fp16 mode: the FFN pair on fp16 operands, as in the full layers (no tape)
            x1 = hbuf = pre = None
            x1h = self._buf(M, d, dev, torch.float16)
            ops.layernorm_fwd(s1, W["g1"], W["b1"], x1h, mean1, rstd1, M, cfg.eps, out32=x1_32)
            hh = self._buf(M, f, dev, torch.float16)
            ops.gemm_nt(x1h, W16["W1"], hh, M, bias=W["bf1"], act=1)
            s2 = self._buf(M, d, dev, sdt)
            ops.gemm_nt(hh, W16["W2"], s2, M, bias=W["bf2"], residual=x1_32, dropout_p=p_h, seed=s_l + 3)
        else:
            x1 = self._buf(M, d, dev, dt16)
            ops.layernorm_fwd(s1, W["g1"], W["b1"], x1, mean1, rstd1, M, cfg.eps, out32=x1_32)
            hbuf = self._buf(M, f, dev, dt16)
            pre = self._buf(M, f, dev, dt16) if save else None
            ops.gemm_nt(x1, W["W1"], hbuf, M, bias=W["bf1"], preact=pre, act=3 if save else 1)
            s2 = self._buf(M, d, dev, sdt)
            ops.gemm_nt(hbuf, W["W2"], s2, M, bias=W["bf2"], residual=x1_32, dropout_p=p_h, seed=s_l + 3)
        yield
        xo = self._buf(M, d, dev, dt16)
        mean2, rstd2 = (torch.empty(M, **f32), torch.empty(M, **f32)) if save else (None, None)
        ops.layernorm_fwd(s2, W["g2"], W["b2"], xo, mean2, rstd2, M, cfg.eps, cls, 1)
        if save:
            tape.layers.append(dict(cls_only=True, x_in=x, kv=kv, xc=xc, qc=qc, ctx=ctxc, probs=probs, s1=s1, mean1=mean1,
                                    rstd1=rstd1, x1=x1, pre=pre, h=hbuf, s2=s2, mean2=mean2, rstd2=rstd2, seed=s_l, p_h=p_h,
                                    p_a=p_a, p_out=p_out))

    def _last_layer_cls_bwd(self, i, a, tape, dcls, partial):
        """Backward of :meth:`_last_layer_cls_fwd`; returns dL/d(layer input) as a full [T, d] tensor (fp32 with the fp32 gradient
        stream, else bf16)."""
        cfg = self.cfg
        d, f, H = cfg.dim, cfg.hidden_dim, cfg.n_heads
        M, L, T = tape.M, tape.L, tape.T
        dev = self.flat_p.device
        W, G = self._layer_weights(i), self._layer_grads(i)
        s_l, p_h, p_a, p_out = a["seed"], a["p_h"], a["p_a"], a["p_out"]
        sdt = torch.float32                                             # this layer's M-row gradient stream stays fp32 in both modes
        bdt = a["h"].dtype                                              # the backward's 16-bit format = the tape's (fp16 or bf16)
        buf = lambda r, c, dv, dt=None: self._buf(r, c, dv, bdt if dt is None else dt)
        gc = dcls.contiguous()                                          # dL/dCLS is the stream's first tensor: no rounding at all
        ds2 = buf(M, d, dev, sdt)
        ds2m = buf(M, d, dev)                                    # 16-bit MFMA operand of the next data-gradient GEMM
        lnq = getattr(self, "_lnq", None)
        f32 = dict(dtype=torch.float32, device=dev)
        own = (lambda: torch.empty(ops.ln_partial_elems(M, d), **f32)) if lnq is not None else (lambda: partial)
        ops.layernorm_bwd(gc, a["s2"], a["mean2"], a["rstd2"], W["g2"], ds2, ds2m, G["g2"], G["b2"], G["bf2"], own(), M, p_h, s_l + 3,
                          accumulate=self._acc, defer=lnq)
        yield
        dF = ds2m if ds2m is not None else ds2
        self._wq.add(dF, a["h"], G["W2"], M)
        dpre = buf(M, f, dev)
        ops.gemm_nt(dF, self.ht(i, "f2"), dpre, M, gelu_pre=a["pre"], act=2)
        self._wq.add(dpre, a["x1"], G["W1"], M, dbias=G["bf1"])
        dx1 = buf(M, d, dev, sdt)
        ops.gemm_nt(dpre, self.ht(i, "f1"), dx1, M, residual=ds2)
        yield
        ds1 = buf(M, d, dev, sdt)
        ds1m = buf(M, d, dev)
        ops.layernorm_bwd(dx1, a["s1"], a["mean1"], a["rstd1"], W["g1"], ds1, ds1m, G["g1"], G["b1"], G["bo"], own(), M, p_out, s_l + 2,
                          accumulate=self._acc, defer=lnq)
        dA = ds1m if ds1m is not None else ds1
        self._wq.add(dA, a["ctx"], G["Wo"], M)
        dctx = buf(M, d, dev)
        ops.gemm_nt(dA, self.ht(i, "o"), dctx, M)
        yield
        dqc = buf(M, d, dev)
        pk = tape.pack
        dkv = buf(T, 2 * d, dev)                                    # (T = the packed row count of a packed batch)
        ops.attention_cls_bwd(a["qc"], a["kv"], a["probs"], dctx, dqc, dkv, M, L, H, p_a, s_l + 1, cu=pk.cu if pk is not None else None)
        self._wq.add(dqc, a["xc"], G["Wqkv"][:d], M, dbias=G["bqkv"][:d])
        self._wq.add(dkv, a["x_in"], G["Wqkv"][d:], T, dbias=G["bqkv"][d:])
        yield
        wt = self.ht(i, "qkv")                                      # [d, 3d] = Wqkv^T
        # the [T, d] gradient this layer hands down: fp32 - or fp16 with the fp16 gradient stream (the layer's own M-row stream stays fp32:
        # the CLS rows are added from the fp32 gq with one rounding)
        g = buf(T, d, dev, torch.float16 if (self.grad_stream16 and bdt == torch.float16) else sdt)
        ops.gemm_nt(dkv, wt[:, d:], g, T)                           # through K and V: every token
        gq = buf(M, d, dev, sdt)
        ops.gemm_nt(dqc, wt[:, :d], gq, M, residual=ds1)            # through Q and the residual: CLS rows only
        if pk is None:
            ops.add_rows_strided(g, gq, M, L)
        else:
            ops.add_rows_idx(g, gq, pk.cls_idx, M)
        return g

    # ------------------------------------------------------------------ backward
    def backward_from_cls(self, tape: _Tape, dcls: torch.Tensor, after_layer=None, accumulate: bool = True, check_grads: bool = False,
                          before_last_wgrad=None):
        """Accumulate parameter gradients of this tower into ``flat_g`` given dL/dCLS (fp32 [M, d]).

        ``after_layer(i)`` (optional) is called when the gradients of transformer layer i are complete (and with
        -1 after the embedding gradients): the hook the trainer uses to launch that bucket's all-reduce.

        ``accumulate=False`` (trainer, unshared towers): every weight / bias / LayerNorm gradient is WRITTEN instead of added
        to, so ``flat_g`` needs no zeroing except the embedding tables (scatter-add by atomics).

        ``before_last_wgrad()`` (optional) is called once, right before the LAST group of deferred weight gradients is launched:
        from there on this stream runs one long launch that is on nobody's critical path."""
        return self.backward_steps(tape, dcls, after_layer=after_layer, accumulate=accumulate, check_grads=check_grads,
                                   before_last_wgrad=before_last_wgrad).finish()

    def backward_steps(self, tape: _Tape, dcls: torch.Tensor, after_layer=None, accumulate: bool = True, check_grads: bool = False,
                       before_last_wgrad=None, window=None) -> Stepper:
        """:meth:`backward_from_cls` as a :class:`Stepper`; ``window(kind)`` as in :meth:`encode_steps` (before LayerNorm / attention backward
        launches)."""
        contexts = []
        base = getattr(self, "seed_base_ptr", None) if getattr(tape, "device_seed", False) else None
        if base:
            contexts.append(lambda: ops.seed_base(base))
        if (self.amp16 and getattr(ops._TLS, "loss_scale", None) is None and tape.layers and tape.layers[-1] is not None
                and tape.layers[-1]["h"].dtype == torch.float16):
            # an fp16 tape outside the trainer (the autograd bridge of the reference-style loop: dL/dCLS arrives unscaled): this tower scales
            # it by its own data-derived power of two (ops.loss_scale_adapt); parameter gradients come out unscaled
            st = self.__dict__.get("_own_scale")
            if st is None or st.device != self.flat_p.device:
                st = self.__dict__["_own_scale"] = ops.new_loss_scale_state(self.flat_p.device)
            dcls = dcls.contiguous().clone()
            ops.loss_scale_adapt(dcls, None, st)
            contexts.append(lambda: ops.loss_scale(st.data_ptr()))
        return Stepper(self._backward_gen(tape, dcls, after_layer, accumulate, check_grads, before_last_wgrad, window), contexts)

    def _backward_gen(self, tape, dcls, after_layer, accumulate, check_grads, before_last_wgrad, window):
        self._acc = bool(accumulate)
        cfg = self.cfg
        self.ensure_grads(check_all=check_grads)
        # Weight gradients are deferred (hip_ops.WgradQueue): only the data gradients are on the critical path.  They are launched
        # as one group at the end of the backward, or every `wgrad_flush_layers` layers when somebody (the trainer's all-reduce
        # hooks) wants layers to complete early; `after_layer(i)` is only called once layer i's weight gradients have been launched.
        self._wq = ops.WgradQueue()
        # LayerNorm gamma / beta (and the preceding Linear's bias) gradients can be deferred likewise (CLDRD_LN_DEFER=1): each
        # layernorm_bwd leaves its per-block sums in a scratch buffer of its own and one launch next to the weight-gradient group reduces
        # them all.  Bit-identical and 22 launches fewer per step.  Round 2 kept it off (±0 at cfg2, slower on the then enqueue-bound
        # cfg1); measured again in round 3 with the step replayed as a HIP graph: -1.0 % step time at cfg2, twice on one box
        # (profiles/r03_microbench.txt): on by default, CLDRD_LN_DEFER=0 restores the immediate reductions.
        self._lnq = ops.LnReduceQueue() if self.ln_defer else None
        flush_every = int(getattr(self, "wgrad_flush_layers", 0) or 0)
        waiting = []

        def layer_done(i, force=False):
            if i == -1 and after_layer is not None:
                # The embedding block (i = -1) is complete as soon as embed_ln_bwd has been launched - it has no deferred weight
                # gradient - so its hook runs BEFORE the last weight-gradient group is launched: the largest all-reduce bucket of the
                # tower (94 MB at DistilBERT) then travels under that group instead of after it.
                after_layer(-1)
            else:
                waiting.append(i)
            # (layer 0 never triggers an early flush: its group waits for the forced one behind embed_ln_bwd, so that the embedding block's
            # hook - the tower's largest all-reduce bucket - is issued IN FRONT of a weight-gradient group it can travel under.  Until round 5
            # a layer count that `wgrad_flush_layers` divides (6 / 3, 12 / 6: every shipped config) launched the last group at layer 0 and the
            # embedding bucket went out with nothing left to overlap it: tools/bucket_plan.py)
            if force or (flush_every > 0 and len(waiting) >= flush_every and i != 0):
                if force and before_last_wgrad is not None:
                    before_last_wgrad()
                # `norm_sink` (the trainer's, one GPU): when this ONE flush produces every layer gradient of the tower - LayerNorm reductions
                # deferred, no early flushes - the two launches also leave the clip norm's partial sums of what they write (hip_ops.norm_sink)
                sink = getattr(self, "norm_sink", None) if (force and flush_every == 0 and self._lnq is not None and not self._acc) else None
                with ops.norm_sink(sink) as ns:
                    if self._lnq is not None:
                        self._lnq.flush(accumulate=self._acc)
                    self._wq.flush(accumulate=self._acc)
                self.norm_sink_used = ns.used if sink is not None else -1
                if after_layer is not None:
                    for j in waiting:
                        after_layer(j)
                waiting.clear()
        M, L, T = tape.M, tape.L, tape.T
        d, f, H = cfg.dim, cfg.hidden_dim, cfg.n_heads
        dev = self.flat_p.device
        f32 = dict(dtype=torch.float32, device=dev)
        partial = torch.empty(max(ops.ln_partial_elems(T, d), ((T + 127) // 128) * max(3 * d, f)), **f32)
        g = gb = None         # gradient of the current layer's output: g (+ gb, a bf16 branch term, with the fp32 gradient stream)
        for i in reversed(range(cfg.n_layers)):
            W, G, a = self._layer_weights(i), self._layer_grads(i), tape.layers[i]
            if a.get("cls_only"):
                g = yield from self._last_layer_cls_bwd(i, a, tape, dcls, partial)
                tape.layers[i] = None
                layer_done(i)
                yield
                continue
            bdt = a["h"].dtype                                          # 16-bit format of this backward = the tape's (fp16 in amp16)
            # the gradient stream (residual path): its arithmetic is fp32 inside the LayerNorm backward, where stream and branch are added;
            # BETWEEN kernels it is stored in fp16 in the fp16 mode (round 5, see __init__), in fp32 in the bf16 mode
            G16 = self.grad_stream16 and bdt == torch.float16
            sdt = torch.float16 if G16 else torch.float32
            buf = lambda r, c, dv, dt=None: self._buf(r, c, dv, bdt if dt is None else dt)
            if g is None:
                g = buf(T, d, dev, sdt)
                if tape.pack is None:
                    ops.scatter_cls_grad(dcls.contiguous(), g, M, L, T)
                else:
                    ops.scatter_cls_grad_idx(dcls.contiguous(), g, tape.pack.cls_idx, T)
            s_l, p_h, p_a, p_out = a["seed"], a["p_h"], a["p_a"], a["p_out"]
            # --- output LayerNorm + FFN ---
            ds2 = buf(T, d, dev, sdt)
            # the 16-bit MFMA operand of the FFN2 data / weight gradients: a tensor of its own when dropout separates it from the stream, or
            # when the stream is fp32; with the fp16 stream and no dropout the stream tensor serves
            ds2m = buf(T, d, dev) if (p_h > 0 or not G16) else None
            lnq = self._lnq
            own = (lambda: torch.empty(ops.ln_partial_elems(T, d), **f32)) if lnq is not None else (lambda: partial)
            # fp32 stream: the gradient of a LayerNorm output is `g` (fp32: the residual path) + `gb` (bf16: the plain output of the
            # branch's last data-gradient GEMM), added inside layernorm_bwd - not in that GEMM's epilogue (200 MB less per GEMM)
            if window is not None:
                window("ln")
            ops.layernorm_bwd(g, a["s2"], a["mean2"], a["rstd2"], W["g2"], ds2, ds2m, G["g2"], G["b2"], G["bf2"], own(), T,
                              p_h, s_l + 3, accumulate=self._acc, defer=lnq, dy_branch=gb)
            yield
            dF = ds2m if ds2m is not None else ds2
            self._wq.add(dF, a["h"], G["W2"], T)
            dpre = buf(T, f, dev)
            ops.gemm_nt(dF, self.ht(i, "f2"), dpre, T, gelu_pre=a["pre"], act=2)
            yield
            self._wq.add(dpre, a["x1"], G["W1"], T, dbias=G["bf1"])
            dx1 = buf(T, d, dev)
            ops.gemm_nt(dpre, self.ht(i, "f1"), dx1, T)                 # the FFN branch alone; the residual path is ds2
            # --- attention-output LayerNorm + attention ---
            ds1 = buf(T, d, dev, sdt)
            ds1m = buf(T, d, dev) if (p_out > 0 or not G16) else None
            yield
            if window is not None:
                window("ln")
            ops.layernorm_bwd(ds2, a["s1"], a["mean1"], a["rstd1"], W["g1"], ds1, ds1m, G["g1"], G["b1"], G["bo"], own(), T,
                              p_out, s_l + 2, accumulate=self._acc, defer=lnq, dy_branch=dx1)
            dA = ds1m if ds1m is not None else ds1
            self._wq.add(dA, a["ctx"], G["Wo"], T)
            dctx = buf(T, d, dev)
            yield
            ops.gemm_nt(dA, self.ht(i, "o"), dctx, T)
            pk = tape.pack
            yield
            if window is not None:
                window("attn")
            dqkv = buf(T, 3 * d, dev)
            for sl, tile in (pk.groups if pk is not None else [(None, 0)]):
                ops.attention_bwd(a["qkv"], tape.mask if pk is None else None, a["ctx"], dctx, a["lse"], dqkv, M, L, H, p_a, s_l + 1,
                                  drop_bits=a.get("dbits"), cu=pk.cu if pk is not None else None, seq_list=sl, tile=tile)
            self._wq.add(dqkv, a["x_in"], G["Wqkv"], T, dbias=G["bqkv"])
            yield
            gb = buf(T, d, dev)
            ops.gemm_nt(dqkv, self.ht(i, "qkv"), gb, T)                 # the attention branch alone
            g = ds1                                                     # the residual path
            tape.layers[i] = None        # this layer's activations: the deferred weight-gradient jobs keep what they still need
            layer_done(i)
            yield
        type0 = self.w("embeddings.token_type_embeddings.weight")[0] if cfg.arch == "bert" else None
        dtype0 = self.g("embeddings.token_type_embeddings.weight")[0] if cfg.arch == "bert" else None
        if window is not None:
            window("ln")
        ops.embed_ln_bwd(g, tape.ids.view(-1), self.w("embeddings.word_embeddings.weight"),
                         self.w("embeddings.position_embeddings.weight"), type0, self.w("embeddings.LayerNorm.weight"),
                         tape.mean0, tape.rstd0, self.g("embeddings.word_embeddings.weight"),
                         self.g("embeddings.position_embeddings.weight"), dtype0, self.g("embeddings.LayerNorm.weight"),
                         self.g("embeddings.LayerNorm.bias"), partial, T, L, tape.p_embed, tape.seed, accumulate=self._acc,
                         pos_idx=tape.pack.pos if tape.pack is not None else None, dy_branch=gb)
        yield
        layer_done(-1, force=True)

    # ------------------------------------------------------------------ HF-style call surface
    def forward(self, input_ids=None, attention_mask=None, **_):
        """``encoder(**enc)[0][:, 0, :]`` compatibility: returns a 1-tuple whose element supports ``[:, 0, :]``."""
        return (_ClsOnly(encode_autograd(self, input_ids, attention_mask)),)

    # ------------------------------------------------------------------ (de)serialisation
    def save_pretrained(self, path: str):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as fh:
            json.dump(self.cfg.to_hf_dict(), fh, indent=1)
        torch.save({k: v.detach().cpu().clone() for k, v in self.state_dict().items()}, os.path.join(path, "pytorch_model.bin"))

    @classmethod
    def from_pretrained(cls, name_or_path, seed: int | None = None, allow_random_init: bool = False) -> "HipEncoder":
        if isinstance(name_or_path, EncoderConfig):
            return cls(name_or_path, seed=seed)
        if os.path.isdir(str(name_or_path)):
            with open(os.path.join(name_or_path, "config.json")) as fh:
                cfg = EncoderConfig.from_hf_dict(json.load(fh))
            enc = cls(cfg, seed=seed)
            st = os.path.join(name_or_path, "model.safetensors")
            if os.path.exists(st):
                from safetensors.torch import load_file
                sd = load_file(st)
            else:
                sd = torch.load(os.path.join(name_or_path, "pytorch_model.bin"), map_location="cpu")
            enc.load_hf_state_dict(sd)
            return enc
        # a hub name: the reference calls AutoModel.from_pretrained(name), which resolves through the local HF cache first
        local = None
        try:
            from huggingface_hub import snapshot_download
            local = snapshot_download(str(name_or_path), local_files_only=True)
        except Exception:
            local = None
        if local is not None and os.path.exists(os.path.join(local, "config.json")):
            return cls.from_pretrained(local, seed=seed)
        if str(name_or_path) in _KNOWN and (allow_random_init or os.environ.get("CLDRD_ALLOW_RANDOM_INIT", "") == "1"):
            # explicit opt-in only (synthetic benchmarks / tests without network): architecture of the named checkpoint, seeded
            # random weights.  Silently distilling into a random student instead of TAS-B would be a wrong result, not a fallback.
            import warnings
            warnings.warn(f"{name_or_path}: pretrained weights not found in the local HF cache; seeded RANDOM init (opt-in)")
            return cls(EncoderConfig(**_KNOWN[str(name_or_path)]), seed=seed)
        raise FileNotFoundError(
            f"{name_or_path}: not a local model directory and not in the local HuggingFace cache (no network here). Pass a model "
            f"directory, an EncoderConfig, or opt in to random weights with CLDRD_ALLOW_RANDOM_INIT=1 / allow_random_init=True.")

    def load_hf_state_dict(self, sd: dict):
        """Accepts HF DistilBertModel / BertModel keys (optional ``distilbert.`` / ``bert.`` prefix; pooler ignored)."""
        own = dict(self.named_parameters())
        seen = set()
        with torch.no_grad():
            for k, v in sd.items():
                kk = k
                for pre in ("distilbert.", "bert."):
                    if kk.startswith(pre):
                        kk = kk[len(pre):]
                if kk in own:
                    own[kk].copy_(v.to(torch.float32))
                    seen.add(kk)
        missing = [n for n in self._names if n not in seen]
        if missing:
            raise KeyError(f"missing keys in checkpoint: {missing[:4]}{'...' if len(missing) > 4 else ''}")


class _ClsOnly:
    """What ``encoder(**enc)[0]`` returns: only the CLS row is materialised; ``[:, 0, :]`` / ``[:, 0]``
    give the fp32 CLS embeddings, anything else is refused loudly."""

    def __init__(self, cls):
        self.cls = cls

    def __getitem__(self, idx):
        if isinstance(idx, tuple) and len(idx) >= 2 and idx[0] == slice(None) and idx[1] == 0 and \
                all(i == slice(None) for i in idx[2:]):
            return self.cls
        raise NotImplementedError("HipEncoder only materialises the CLS token: use last_hidden_state[:, 0, :]")


class _EncodeFn(torch.autograd.Function):
    """Autograd bridge for the reference-style loop (``loss.backward()``): the tower's hand-written backward runs
    when the CLS gradient arrives; parameter gradients are accumulated straight into ``flat_g`` / ``param.grad``."""

    @staticmethod
    def forward(ctx, anchor, enc, ids, mask, fp16, lengths=None):
        cls, tape = enc.encode(ids, mask, save=True, fp16=fp16, lengths=lengths)
        ctx.enc, ctx.tape = enc, tape
        return cls

    @staticmethod
    def backward(ctx, dcls):
        ctx.enc.backward_from_cls(ctx.tape, dcls.contiguous().float(), check_grads=True)
        ctx.tape = None
        return None, None, None, None, None, None


def encode_autograd(enc: HipEncoder, ids, mask, fp16=None, lengths=None):
    if torch.is_grad_enabled() and any(p.requires_grad for p in enc.parameters()):
        if getattr(enc, "_anchor", None) is None or enc._anchor.device != enc.flat_p.device:
            enc._anchor = torch.zeros(1, device=enc.flat_p.device, requires_grad=True)
        return _EncodeFn.apply(enc._anchor, enc, ids, mask, fp16, lengths)
    return enc.encode(ids, mask, save=False, fp16=fp16, lengths=lengths)
