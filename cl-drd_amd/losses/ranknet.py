import torch

from ._fn import loss_value


def ranknet_loss(y_pred, y_true, eps=1e-10, padded_value_indicator=-1, reduction="mean", sigma=1.):
    """reference losses/ranknet.py:3-44.  y_pred, y_true: FloatTensor [bz, topk].  ``eps`` and ``sigma`` are unused
    there too; padded labels are asserted absent (``ranknet.py:16``, one host sync, as in the reference)."""
    assert torch.sum(y_true == padded_value_indicator) == 0
    return loss_value(y_pred, y_true, "ranknet", pad=padded_value_indicator, reduction=reduction)
