import torch.nn as nn

from ._fn import loss_value


class KLDiv(nn.Module):
    """reference losses/kl_div.py:5-22: KLDivLoss(batchmean)(log_softmax(y_pred / T), softmax(y_true / T)) (no T^2 factor)."""

    def __init__(self, T=1.):
        super(KLDiv, self).__init__()
        self.T = T

    def forward(self, y_pred, y_true):
        assert y_pred.dim() == y_true.dim() == 2
        return loss_value(y_pred, y_true, "kl_div", T=self.T)
