"""Oracle (TEST INFRASTRUCTURE): trainer-step arithmetic after ``backward``.

Follows ``trainer/multistep-curriculum/nway_listwise_1.py:258-266,353-367``:
``clip_grad_norm_(max_grad_norm)`` -> legacy ``transformers.AdamW`` step -> linear warmup/decay.

PARITY UNPINNED for the AdamW step: ``transformers.AdamW`` no longer exists in the installed
transformers 5.15, so this restates its published update rule (``correct_bias=True``: bias correction
folded into the step size, eps added to sqrt(v) un-corrected, decoupled weight decay applied *after*
the Adam update with the scheduled lr).  The LR schedule *is* pinned against
``transformers.get_linear_schedule_with_warmup`` by ``tests/golden/lr_schedule.npz``.
"""
from __future__ import annotations

import math

import numpy as np


def linear_schedule_factor(step: int, warmup_steps: int, total_steps: int) -> float:
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    return max(0.0, float(total_steps - step) / float(max(1, total_steps - warmup_steps)))


def clip_coef(grads, max_norm: float) -> tuple[float, float]:
    """(total L2 norm, multiplier) of ``torch.nn.utils.clip_grad_norm_``: coef = min(1, max_norm / (norm + 1e-6))."""
    total = math.sqrt(sum(float((np.asarray(g, dtype=np.float64) ** 2).sum()) for g in grads))
    return total, min(1.0, max_norm / (total + 1e-6))


def adamw_step(p, g, m, v, *, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """One legacy-HF-AdamW update in float64; ``step`` is 1-based.  Returns new (p, m, v)."""
    p, g, m, v = (np.asarray(a, dtype=np.float64) for a in (p, g, m, v))
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    step_size = lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    p = p - step_size * m / (np.sqrt(v) + eps)
    if weight_decay > 0.0:
        p = p - lr * weight_decay * p
    return p, m, v


def no_decay(name: str) -> bool:
    """Reference trainer/multistep-curriculum/nway_listwise_1.py:259-263: names containing ``bias`` or
    ``LayerNorm.weight`` are not decayed (DistilBERT's ``sa_layer_norm`` / ``output_layer_norm``
    weights do not match and ARE decayed)."""
    return ("bias" in name) or ("LayerNorm.weight" in name)
