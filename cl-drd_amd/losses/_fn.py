from __future__ import annotations

import torch

from .. import hip_ops as ops


class _LossFn(torch.autograd.Function):
    """value + gradient in one launch; backward only scales the stored gradient by the incoming scalar."""

    @staticmethod
    def forward(ctx, y_pred, y_true, kind, batch_weight, T, pad, reduction):
        out, grad = ops.loss_fwd_bwd(kind, y_pred.detach().float(), y_true.detach().float(), batch_weight=batch_weight,
                                     T=T, pad_indicator=pad, reduction=reduction)
        ctx.save_for_backward(grad)
        ctx.in_dtype = y_pred.dtype
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g).to(ctx.in_dtype), None, None, None, None, None, None


def loss_value(y_pred, y_true, kind, *, batch_weight=None, T=1.0, pad=-1.0, reduction="mean"):
    if reduction not in ("mean", "sum"):
        raise ValueError("Reduction method can be either sum or mean")
    if not y_pred.is_cuda:
        raise RuntimeError("cldrd_amd.losses run on the GPU only (no CPU path)")
    bw = None if batch_weight is None else batch_weight.detach().float().contiguous()
    return _LossFn.apply(y_pred, y_true, kind, bw, float(T), float(pad), reduction)
