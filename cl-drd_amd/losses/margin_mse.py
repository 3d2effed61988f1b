import torch.nn as nn

from ._fn import loss_value


class MarginMSE(nn.Module):
    """reference losses/margin_mse.py:4-19: mean over all ordered pairs (i, j) of ((s_i - s_j) - (t_i - t_j))^2."""

    def __init__(self):
        super(MarginMSE, self).__init__()

    def forward(self, M_s, M_t):
        assert M_s.dim() == M_t.dim() == 2
        return loss_value(M_s, M_t, "margin_mse")
