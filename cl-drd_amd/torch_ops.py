"""``torch.ops.cldrd.*``: the operators of the hot path registered with PyTorch's dispatcher (``torch.library``).

The reference has no operator / plugin interface of its own - its hot path bottoms out in torch, transformers and faiss calls - so
the drop-in boundary is the Python call surface (``models.nway_dual_encoder``, ``losses.*``, ``retriever.*``) and, below it, this
operator namespace over the one hipcc-built library.  Every op here is a thin functional wrapper around a
C-ABI entry point of ``include/cldrd_hip.h`` (through ``hip_ops``): CUDA (= HIP) kernels only - a CPU tensor finds no kernel and
raises - plus a fake (meta) implementation for shape inference under ``torch.compile`` / ``FakeTensorMode``, and an autograd
formula where the reference differentiates through the call:

    cldrd::listwise_loss      losses/{kl_div,margin_mse,ranknet,lambda_rank,weighted_pointwise}.py     value + d loss / d y_pred in one launch
    cldrd::lambda_loss        losses/standard_lambda_rank.py:3-95
    cldrd::nway_score         models/nway_dual_encoder.py:30-47 (N-way / in-batch scoring), backward cldrd::nway_score_bwd
    cldrd::linear             torch.nn.Linear inside the HF encoder (bias / erf-GELU / residual epilogue), forward
    cldrd::layer_norm         HF LayerNorm (eps 1e-12), forward
    cldrd::self_attention     HF DistilBertSelfAttention / BertSelfAttention (head dim 64, L <= 256), forward

The fused trainer (``trainer.nway_listwise.NwayTrainer``) keeps calling the C ABI directly: 240 launches a step do not go through
the dispatcher one by one.  ``losses.*`` and ``NwayDualEncoder.forward`` (the reference-style loop) go through these ops.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from . import hip_ops as ops

_KINDS = ("kl_div", "margin_mse", "ranknet", "lambda_mrr", "weighted_pointwise")
_SCHEMES = (None, "ndcgLoss1_scheme", "ndcgLoss2_scheme", "lambdaRank_scheme", "ndcgLoss2PP_scheme", "rankNet_scheme",
            "rankNetWeightedByGTDiff_scheme", "rankNetWeightedByGTDiffPowed_scheme")


# ---------------------------------------------------------------------------------------------------------------- losses
@torch.library.custom_op("cldrd::listwise_loss", mutates_args=(), device_types="cuda")
def listwise_loss(y_pred: Tensor, y_true: Tensor, kind: int, batch_weight: Optional[Tensor], T: float, pad: float,
                  mean_reduction: bool) -> Tuple[Tensor, Tensor]:
    """(loss_out float[2] = {loss, valid pair count}, d loss / d y_pred [B, N]).  kind indexes
    (kl_div, margin_mse, ranknet, lambda_mrr, weighted_pointwise); batch_weight != None with lambda_mrr = bweight_lambda_mrr_loss."""
    out, grad = ops.loss_fwd_bwd(_KINDS[kind], y_pred.detach().float(), y_true.detach().float(), batch_weight=batch_weight, T=T,
                                 pad_indicator=pad, reduction="mean" if mean_reduction else "sum")
    return out, grad


@listwise_loss.register_fake
def _(y_pred, y_true, kind, batch_weight, T, pad, mean_reduction):
    return y_pred.new_empty((2,), dtype=torch.float32), y_pred.new_empty(y_pred.shape, dtype=torch.float32)


def _loss_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])
    ctx.in_dtype = inputs[0].dtype


def _loss_backward(ctx, g_out, g_grad):
    (grad,) = ctx.saved_tensors
    return (grad * g_out[0]).to(ctx.in_dtype), None, None, None, None, None, None


torch.library.register_autograd("cldrd::listwise_loss", _loss_backward, setup_context=_loss_setup)


@torch.library.custom_op("cldrd::lambda_loss", mutates_args=(), device_types="cuda")
def lambda_loss(y_pred: Tensor, y_true: Tensor, scheme: int, k: int, eps: float, sigma: float, mu: float, pad: float,
                mean_reduction: bool, log2_reduction: bool, gain_linear: bool) -> Tuple[Tensor, Tensor]:
    """allRank LambdaLoss (losses/standard_lambda_rank.py:3-95): scheme indexes (None, ndcgLoss1, ndcgLoss2, lambdaRank, ndcgLoss2PP,
    rankNet, rankNetWeightedByGTDiff, rankNetWeightedByGTDiffPowed); k <= 0: no truncation."""
    out, grad = ops.lambda_loss_fwd_bwd(y_pred.detach().float(), y_true.detach().float(), eps=eps, padded_value_indicator=pad,
                                        weighing_scheme=_SCHEMES[scheme], k=(k if k > 0 else None), sigma=sigma, mu=mu,
                                        reduction="mean" if mean_reduction else "sum",
                                        reduction_log="binary" if log2_reduction else "natural", gain="linear" if gain_linear else "power")
    return out, grad


@lambda_loss.register_fake
def _(y_pred, y_true, scheme, k, eps, sigma, mu, pad, mean_reduction, log2_reduction, gain_linear):
    return y_pred.new_empty((2,), dtype=torch.float32), y_pred.new_empty(y_pred.shape, dtype=torch.float32)


def _lambda_backward(ctx, g_out, g_grad):
    (grad,) = ctx.saved_tensors
    return ((grad * g_out[0]).to(ctx.in_dtype),) + (None,) * 10


torch.library.register_autograd("cldrd::lambda_loss", _lambda_backward, setup_context=_loss_setup)


# ---------------------------------------------------------------------------------------------------------------- scoring
def _score_cols(B: int, N: int, mode: int) -> int:
    return N if mode == 0 else (B * N if mode == 1 else 2 * N)


@torch.library.custom_op("cldrd::nway_score", mutates_args=(), device_types="cuda")
def nway_score(q: Tensor, p: Tensor, B: int, N: int, mode: int) -> Tensor:
    """logits[B, N'] of models/nway_dual_encoder.py:30-47: q [B, D], p [B*N, D] fp32; mode 0 N-way, 1 all in-batch negatives
    [B, B*N], 2 next sample's N [B, 2N]."""
    q, p = q.contiguous(), p.contiguous()
    logits = torch.empty(B, _score_cols(B, N, mode), dtype=torch.float32, device=q.device)
    ops.score_fwd(q, p, logits, B, N, mode)
    return logits


@nway_score.register_fake
def _(q, p, B, N, mode):
    return q.new_empty((B, _score_cols(B, N, mode)), dtype=torch.float32)


@torch.library.custom_op("cldrd::nway_score_bwd", mutates_args=(), device_types="cuda")
def nway_score_bwd(dlogits: Tensor, q: Tensor, p: Tensor, B: int, N: int, mode: int) -> Tuple[Tensor, Tensor]:
    # row-major outputs whatever the strides of q / p (empty_like would keep the strides of a dense non-contiguous input, e.g. a
    # transposed view, while the kernel writes row-major: silently permuted gradients)
    dq = torch.empty(q.shape, dtype=q.dtype, device=q.device)
    dp = torch.empty(p.shape, dtype=p.dtype, device=p.device)
    ops.score_bwd(dlogits.contiguous().float(), q.contiguous(), p.contiguous(), dq, dp, B, N, mode)
    return dq, dp


@nway_score_bwd.register_fake
def _(dlogits, q, p, B, N, mode):
    return q.new_empty(q.shape), p.new_empty(p.shape)


def _score_setup(ctx, inputs, output):
    q, p, B, N, mode = inputs
    ctx.save_for_backward(q, p)
    ctx.dims = (B, N, mode)


def _score_backward(ctx, dlogits):
    q, p = ctx.saved_tensors
    dq, dp = torch.ops.cldrd.nway_score_bwd(dlogits, q, p, *ctx.dims)
    return dq, dp, None, None, None


torch.library.register_autograd("cldrd::nway_score", _score_backward, setup_context=_score_setup)


# ------------------------------------------------------------------------------------------- encoder building blocks (forward)
@torch.library.custom_op("cldrd::linear", mutates_args=(), device_types="cuda")
def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor], residual: Optional[Tensor], gelu: bool, out_f32: bool) -> Tensor:
    """epilogue(x[M, K] @ weight[N, K]^T): + bias (fp32 [N]) -> erf-GELU -> + residual (16-bit or fp32 [M, N]); x / weight bf16 or fp16
    (the fp16 format: M < 1024, forward only); output in the operand format, or fp32."""
    out = torch.empty(x.shape[0], weight.shape[0], dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    ops.gemm_nt(x, weight, out, bias=bias, residual=residual, act=1 if gelu else 0)
    return out


@linear.register_fake
def _(x, weight, bias, residual, gelu, out_f32):
    return x.new_empty((x.shape[0], weight.shape[0]), dtype=torch.float32 if out_f32 else x.dtype)


@torch.library.custom_op("cldrd::layer_norm", mutates_args=(), device_types="cuda")
def layer_norm(x: Tensor, weight: Tensor, bias: Tensor, eps: float) -> Tensor:
    """LayerNorm over the last dim of x [T, d] (bf16, or fp32 = the fp32 residual stream); bf16 output; statistics in fp32."""
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    ops.layernorm_fwd(x.contiguous(), weight, bias, out, None, None, x.shape[0], eps)
    return out


@layer_norm.register_fake
def _(x, weight, bias, eps):
    return x.new_empty(x.shape, dtype=torch.bfloat16)


@torch.library.custom_op("cldrd::self_attention", mutates_args=(), device_types="cuda")
def self_attention(qkv: Tensor, mask: Optional[Tensor], nseq: int, L: int, H: int) -> Tensor:
    """softmax(Q K^T / 8 + key mask) V for packed qkv [nseq * L, 3 * H * 64] (Q | K | V) -> ctx [nseq * L, H * 64]; mask int64 [nseq, L]."""
    ctx = torch.empty(qkv.shape[0], H * 64, dtype=qkv.dtype, device=qkv.device)
    ops.attention_fwd(qkv.contiguous(), mask, ctx, None, nseq, L, H)
    return ctx


@self_attention.register_fake
def _(qkv, mask, nseq, L, H):
    return qkv.new_empty((qkv.shape[0], H * 64))


OPS = ("listwise_loss", "lambda_loss", "nway_score", "nway_score_bwd", "linear", "layer_norm", "self_attention")
