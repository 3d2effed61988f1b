"""``lambda_loss`` of reference ``losses/standard_lambda_rank.py:3-95`` (allRank's LambdaLoss framework) with its seven
weighing schemes (``:98-127``), on MI355X: value and analytic gradient from one fused kernel (csrc/loss.hip,
``lambda_loss_row_kernel``)."""
import torch

from .. import hip_ops as ops
from .. import torch_ops  # noqa: F401  (registers torch.ops.cldrd.*)

WEIGHING_SCHEMES = tuple(k for k in ops.LAMBDA_SCHEMES if k is not None)


def lambda_loss(y_pred, y_true, eps=1e-4, padded_value_indicator=-1, weighing_scheme=None, k=None, sigma=1., mu=10.,
                reduction="mean", reduction_log="natural", gain="power"):
    """Same arguments as the reference.  y_pred, y_true: [batch_size, slate_length]; ``padded_value_indicator`` marks padded
    items in ``y_true``; ``k`` truncates the loss (and maxDCG) at rank k; ``weighing_scheme`` is one of WEIGHING_SCHEMES or None."""
    if not y_pred.is_cuda:
        raise RuntimeError("cldrd_amd.losses run on the GPU only (no CPU path)")
    if weighing_scheme not in ops.LAMBDA_SCHEMES:
        raise KeyError(weighing_scheme)               # the reference looks the scheme up in globals()
    if gain not in ("power", "linear"):
        raise ValueError(f"{gain} not defined.")
    if reduction_log not in ("natural", "binary"):
        raise ValueError("Reduction logarithm base can be either natural or binary")
    if reduction not in ("mean", "sum"):
        raise ValueError("Reduction method can be either sum or mean")
    out, _ = torch.ops.cldrd.lambda_loss(y_pred, y_true, ops.LAMBDA_SCHEMES[weighing_scheme], 0 if k is None else int(k), float(eps),
                                         float(sigma), float(mu), float(padded_value_indicator), reduction == "mean",
                                         reduction_log == "binary", gain == "linear")
    return out[0]
