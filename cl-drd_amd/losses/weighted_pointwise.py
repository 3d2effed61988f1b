import torch

from ._fn import loss_value


def weighted_pointwise_loss(y_pred, y_weight, T=1.):
    """reference losses/weighted_pointwise.py:3-14: mean of log(1 + exp(-y_pred / T)) * y_weight over [bz, topk + topN];
    negative weights are rejected (one host sync, as in the reference)."""
    assert torch.sum(y_weight < 0) == 0.
    return loss_value(y_pred, y_weight, "weighted_pointwise", T=T)
