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


# ---------------------------------------------------------------------------------------------------------------------
# losses/standard_lambda_rank.py:3-127 (allRank lambda_loss + weighing schemes) and losses/weighted_pointwise.py:3-14
# ---------------------------------------------------------------------------------------------------------------------
LAMBDA_SCHEMES = (None, "ndcgLoss1_scheme", "ndcgLoss2_scheme", "lambdaRank_scheme", "ndcgLoss2PP_scheme", "rankNet_scheme",
                  "rankNetWeightedByGTDiff_scheme", "rankNetWeightedByGTDiffPowed_scheme")


def lambda_loss(y_pred, y_true, eps=1e-4, padded_value_indicator=-1, weighing_scheme=None, k=None, sigma=1.0, mu=10.0,
                reduction="mean", reduction_log="natural", gain="power"):
    """float64 closed form of the reference's lambda_loss: value and d value / d y_pred.

    Slate positions are the stable descending order of the masked predictions; a pair (p, q) of positions counts when both
    are real, both < k, and (except ndcgLoss1_scheme) label[p] > label[q]; its term is -log(max(max(sigmoid(sigma d), eps)^W, eps))
    with d = s_p - s_q clamped to +-1e8; the gradient goes through d only."""
    if weighing_scheme not in LAMBDA_SCHEMES:
        raise KeyError(weighing_scheme)
    s, t = _f64(y_pred), _f64(y_true)
    B, N = s.shape
    kk = N if k is None else min(int(k), N)
    grad = np.zeros_like(s)
    total, count = 0.0, 0
    logk = 1.0 if reduction_log == "natural" else 1.0 / np.log(2.0)
    if reduction_log not in ("natural", "binary"):
        raise ValueError("Reduction logarithm base can be either natural or binary")
    # the reference builds the discounts in float32 whatever the input dtype (standard_lambda_rank.py:50-51:
    # torch.log2(1. + pos_idxs.float())), and the discount differences of the schemes are float32 arithmetic too
    D32 = np.log2((1.0 + np.arange(1, N + 1)).astype(np.float32)).astype(np.float32)
    inv32 = (np.float32(1.0) / D32).astype(np.float32)
    D = D32.astype(np.float64)
    for b in range(B):
        pad = t[b] == padded_value_indicator
        sk = np.where(pad, -np.inf, s[b])
        tk = np.where(pad, -np.inf, t[b])
        order = np.array(sorted(range(N), key=lambda i: (-sk[i], i)))            # stable descending
        ps, ts = sk[order], tk[order]
        tc = np.maximum(ts, 0.0)
        t_sorted = np.maximum(np.sort(tk)[::-1], 0.0)
        if gain == "power":
            maxdcg = max(float(np.sum(((2.0 ** t_sorted - 1.0) / D)[:kk])), eps)
            G = (2.0 ** tc - 1.0) / maxdcg
        elif gain == "linear":
            maxdcg = max(float(np.sum(((t_sorted - 1.0) / D)[:kk])), eps)
            G = (tc - 1.0) / maxdcg
        else:
            raise ValueError(f"{gain} not defined.")

        def weight(p, q):
            lr = float(np.abs(inv32[p] - inv32[q])) * abs(G[p] - G[q])
            dl = abs(p - q)
            n2 = 0.0 if dl == 0 else float(np.abs(inv32[dl - 1] - inv32[dl])) * abs(G[p] - G[q])
            return {None: 1.0, "rankNet_scheme": 1.0, "ndcgLoss1_scheme": G[p] / D[p], "ndcgLoss2_scheme": n2, "lambdaRank_scheme": lr,
                    "ndcgLoss2PP_scheme": mu * n2 + lr, "rankNetWeightedByGTDiff_scheme": abs(tc[p] - tc[q]),
                    "rankNetWeightedByGTDiffPowed_scheme": abs(tc[p] ** 2 - tc[q] ** 2)}[weighing_scheme]

        gp = np.zeros(N)
        for p in range(kk):
            if not np.isfinite(ts[p]):
                continue
            for q in range(kk):
                if not np.isfinite(ts[q]):
                    continue
                if weighing_scheme != "ndcgLoss1_scheme" and not ts[p] - ts[q] > 0:
                    continue
                W = weight(p, q)
                draw = ps[p] - ps[q]
                d = min(max(draw, -1e8), 1e8)
                u = 1.0 / (1.0 + np.exp(-sigma * d))
                a = max(u, eps)
                bw = a ** W
                c = max(bw, eps)
                total += -logk * np.log(c)
                count += 1
                if p != q and bw >= eps and u >= eps and abs(draw) <= 1e8:
                    dt = -logk * (W * bw / a) / c * sigma * u * (1.0 - u)
                    gp[p] += dt
                    gp[q] -= dt
        grad[b, order] = gp
    if reduction == "sum":
        return float(total), grad
    if reduction == "mean":
        return (float(total) / count if count else float("nan")), (grad / count if count else grad * np.nan)
    raise ValueError("Reduction method can be either sum or mean")


def weighted_pointwise(y_pred, y_weight, T: float = 1.0):
    s, w = _f64(y_pred), _f64(y_weight)
    z = -s / T
    val = float(np.mean(np.logaddexp(0.0, z) * w))
    grad = -(1.0 / (1.0 + np.exp(-z))) / T * w / s.size
    return val, grad
