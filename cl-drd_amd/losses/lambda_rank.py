import torch

from ._fn import loss_value


def bweight_lambda_mrr_loss(y_pred, y_true, batch_weight, eps=1e-10, padded_value_indicator=-1, reduction="mean", sigma=1.):
    """reference losses/lambda_rank.py:3-51: lambda_mrr with row b scaled by batch_weight[b]; padding asserted absent (:18)."""
    assert torch.sum(y_true == padded_value_indicator) == 0
    return loss_value(y_pred, y_true, "lambda_mrr", batch_weight=batch_weight, pad=padded_value_indicator, reduction=reduction)


def lambda_mrr_loss(y_pred, y_true, eps=1e-10, padded_value_indicator=-1, reduction="mean", sigma=1.):
    """reference losses/lambda_rank.py:53-96: pairwise logistic loss over pairs with y_true_i > y_true_j, weighted by
    |1/rank_i - 1/rank_j| (ranks of y_pred, non-differentiable); entries equal to ``padded_value_indicator`` are masked."""
    return loss_value(y_pred, y_true, "lambda_mrr", pad=padded_value_indicator, reduction=reduction)
