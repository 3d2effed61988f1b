"""Oracle (TEST INFRASTRUCTURE): CPU restatement of the reference's list-wise distillation losses.

Each function returns ``(value, grad_wrt_y_pred)`` computed in float64 from closed forms, so it is an
independent check of both the reference formulas and the HIP kernels' analytic backward:

  * ``kl_div``        <- ``losses/kl_div.py:11-22``   KLDivLoss(batchmean)(log_softmax(s/T), softmax(t/T))
  * ``margin_mse``    <- ``losses/margin_mse.py:8-19``  mean over all ordered pairs (i, j), i == j included
  * ``ranknet``       <- ``losses/ranknet.py:3-44``   pairwise logistic over pairs with t_i > t_j
  * ``lambda_mrr``    <- ``losses/lambda_rank.py:53-96``  ranknet x |1/rank_i - 1/rank_j| (ranks of y_pred)
  * ``bweight_lambda_mrr`` <- ``losses/lambda_rank.py:3-51``  per-row weight

Rank of item i = 1 + #{j : s_j > s_i or (s_j == s_i and j < i)} (the reference uses ``torch.sort``,
whose tie order is unspecified; tests avoid exact ties).  ``padded_value_indicator`` entries are
excluded from every pair and receive zero gradient; they sort last.
"""
from __future__ import annotations

import numpy as np


def _f64(a):
    return np.asarray(a, dtype=np.float64)


def kl_div(y_pred, y_true, T: float = 1.0):
    s, t = _f64(y_pred) / T, _f64(y_true) / T
    B = s.shape[0]
    ls = s - s.max(1, keepdims=True)
    ls = ls - np.log(np.exp(ls).sum(1, keepdims=True))
    lt = t - t.max(1, keepdims=True)
    lt = lt - np.log(np.exp(lt).sum(1, keepdims=True))
    pt = np.exp(lt)
    val = float((pt * (lt - ls)).sum() / B)
    grad = (np.exp(ls) - pt) / (B * T)
    return val, grad


def margin_mse(M_s, M_t):
    d = _f64(M_s) - _f64(M_t)
    B, N = d.shape
    sd = d.sum(1, keepdims=True)
    val = float(2.0 * (N * (d * d).sum() - (sd * sd).sum()) / (B * N * N))
    grad = 4.0 * (N * d - sd) / (B * N * N)
    return val, grad


def ranks_desc(s: np.ndarray, pad: np.ndarray) -> np.ndarray:
    """1-based rank of every item in the descending order of s; padded items sort last. [B, N] int64."""
    B, N = s.shape
    key = np.where(pad, -np.inf, s)
    gt = key[:, None, :] > key[:, :, None]                       # [b, i, j]: s_j > s_i
    eq = (key[:, None, :] == key[:, :, None]) & (np.arange(N)[None, None, :] < np.arange(N)[None, :, None])
    return 1 + (gt | eq).sum(-1)


def _pairwise(y_pred, y_true, weight_by_rank: bool, batch_weight=None, padded_value_indicator=-1, reduction="mean"):
    if reduction not in ("mean", "sum"):
        raise ValueError("Reduction method can be either sum or mean")
    s, t = _f64(y_pred), _f64(y_true)
    B, N = s.shape
    pad = t == padded_value_indicator
    valid = (~pad[:, :, None]) & (~pad[:, None, :]) & (t[:, :, None] > t[:, None, :])   # pair (i, j), t_i > t_j
    diff = np.clip(s[:, :, None] - s[:, None, :], -1e8, 1e8)
    diff = np.where(valid, diff, 0.0)
    # log(1 + exp(-x)), stable
    loss_ij = np.where(diff > 0, np.log1p(np.exp(-np.abs(diff))), -diff + np.log1p(np.exp(-np.abs(diff))))
    sig = 1.0 / (1.0 + np.exp(diff))                             # sigma(-x) = -d loss / d x
    w = np.ones((B, N, N))
    if weight_by_rank:
        r = ranks_desc(s, pad).astype(np.float64)
        w = np.abs(1.0 / r[:, :, None] - 1.0 / r[:, None, :])
    if batch_weight is not None:
        w = w * _f64(batch_weight).reshape(B, 1, 1)
    w = np.where(valid, w, 0.0)
    cnt = int(valid.sum())
    total = float((w * loss_ij).sum())
    g_pair = -(w * sig)                                          # d/d s_i of pair (i, j); d/d s_j is the negative
    grad = g_pair.sum(2) - g_pair.sum(1)
    if reduction == "mean":
        if cnt == 0:
            return float("nan"), np.full_like(s, np.nan)
        return total / cnt, grad / cnt
    return total, grad


def ranknet(y_pred, y_true, padded_value_indicator=-1, reduction="mean"):
    assert not np.any(_f64(y_true) == padded_value_indicator)   # reference losses/ranknet.py:16
    return _pairwise(y_pred, y_true, False, None, padded_value_indicator, reduction)


def lambda_mrr(y_pred, y_true, padded_value_indicator=-1, reduction="mean"):
    return _pairwise(y_pred, y_true, True, None, padded_value_indicator, reduction)


def bweight_lambda_mrr(y_pred, y_true, batch_weight, padded_value_indicator=-1, reduction="mean"):
    assert not np.any(_f64(y_true) == padded_value_indicator)   # reference losses/lambda_rank.py:18
    return _pairwise(y_pred, y_true, True, batch_weight, padded_value_indicator, reduction)


def train_mrr_recall(logits, labels, topk: int = 10):
    """Per-batch train MRR@k / Recall@k w.r.t. the position of ``labels == 1``
    (reference trainer/multistep-curriculum/nway_listwise_1.py:375-385)."""
    logits, labels = _f64(logits), _f64(labels)
    order = np.argsort(-logits, axis=-1, kind="stable")
    lab = np.take_along_axis(labels, order, axis=-1)
    first = np.where(lab == 1)[1]
    keep = first[first < topk]
    if len(keep) == 0:
        return 0.0, 0.0
    return float(np.sum(1.0 / (keep + 1.0)) / len(first)), float(len(keep) / len(first))
