from __future__ import annotations

import torch

from .. import torch_ops  # noqa: F401  (registers torch.ops.cldrd.*)
from ..torch_ops import _KINDS


def loss_value(y_pred, y_true, kind, *, batch_weight=None, T=1.0, pad=-1.0, reduction="mean"):
    """value + analytic gradient from one fused kernel (``torch.ops.cldrd.listwise_loss``, csrc/loss.hip); the backward only
    scales the stored gradient by the incoming scalar."""
    if reduction not in ("mean", "sum"):
        raise ValueError("Reduction method can be either sum or mean")
    if not y_pred.is_cuda:
        raise RuntimeError("cldrd_amd.losses run on the GPU only (no CPU path)")
    bw = None if batch_weight is None else batch_weight.detach().float().contiguous()
    out, _ = torch.ops.cldrd.listwise_loss(y_pred, y_true, _KINDS.index(kind), bw, float(T), float(pad), reduction == "mean")
    return out[0]
