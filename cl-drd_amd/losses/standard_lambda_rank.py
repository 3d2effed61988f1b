"""``lambda_loss`` of reference ``losses/standard_lambda_rank.py:3-95`` (allRank's LambdaLoss framework) with its seven
weighing schemes (``:98-127``), on MI355X: value and analytic gradient from one fused kernel (csrc/loss.hip,
``lambda_loss_row_kernel``)."""
import torch

from .. import hip_ops as ops

WEIGHING_SCHEMES = tuple(k for k in ops.LAMBDA_SCHEMES if k is not None)


class _LambdaLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y_pred, y_true, kw):
        out, grad = ops.lambda_loss_fwd_bwd(y_pred.detach().float(), y_true.detach().float(), **kw)
        ctx.save_for_backward(grad)
        ctx.in_dtype = y_pred.dtype
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g).to(ctx.in_dtype), None, None


def lambda_loss(y_pred, y_true, eps=1e-4, padded_value_indicator=-1, weighing_scheme=None, k=None, sigma=1., mu=10.,
                reduction="mean", reduction_log="natural", gain="power"):
    """Same arguments as the reference.  y_pred, y_true: [batch_size, slate_length]; ``padded_value_indicator`` marks padded
    items in ``y_true``; ``k`` truncates the loss (and maxDCG) at rank k; ``weighing_scheme`` is one of WEIGHING_SCHEMES or None."""
    if not y_pred.is_cuda:
        raise RuntimeError("cldrd_amd.losses run on the GPU only (no CPU path)")
    kw = dict(eps=eps, padded_value_indicator=padded_value_indicator, weighing_scheme=weighing_scheme, k=k, sigma=sigma, mu=mu,
              reduction=reduction, reduction_log=reduction_log, gain=gain)
    return _LambdaLossFn.apply(y_pred, y_true, kw)
