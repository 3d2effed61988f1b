#!/usr/bin/env python3
"""CPU emulation of the GPU pipeline's rounding points (which tensors are stored in bf16 between kernels) on the full-size
goldens: how much of the bf16 drift each storage decision is worth, relative to the reference's own bf16-autocast drift.
Design aid for the fp32 residual stream; needs no GPU.

    python tools/drift_emulation.py [cfg1|cfg2]
"""
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cldrd_amd.synthetic as syn  # noqa: E402
from oracle import encoder_ref as E  # noqa: E402


def r(t):
    return t.to(torch.bfloat16).float()


def forward(w, cfg, ids, mask, mode, hiddens=None):
    """mode: dict(sum32=bool, res32=bool, grad-free).  Returns CLS fp32 [M, d]; `hiddens` (a list) receives the bf16 input of
    every layer."""
    M, L = ids.shape
    d, H = cfg.dim, cfg.n_heads
    dh = d // H
    x32 = w["embeddings.word_embeddings.weight"][ids] + w["embeddings.position_embeddings.weight"][:L][None]
    if cfg.arch == "bert":
        x32 = x32 + w["embeddings.token_type_embeddings.weight"][0][None, None]
    x32 = F.layer_norm(x32, (d,), w["embeddings.LayerNorm.weight"], w["embeddings.LayerNorm.bias"], cfg.eps)
    xb = r(x32)
    if not mode["res32"]:
        x32 = xb
    bias = torch.zeros(M, 1, 1, L).masked_fill(mask[:, None, None, :] == 0, torch.finfo(torch.float32).min)
    for i in range(cfg.n_layers):
        if hiddens is not None:
            hiddens.append(xb)
        q_n, k_n, v_n, o_n, ln1_n, f1_n, f2_n, ln2_n = E._layer_names(cfg, i)
        lin = lambda a, n: F.linear(a, r(w[n + ".weight"]), w[n + ".bias"])
        q = r(lin(xb, q_n)).view(M, L, H, dh).transpose(1, 2)
        k = r(lin(xb, k_n)).view(M, L, H, dh).transpose(1, 2)
        v = r(lin(xb, v_n)).view(M, L, H, dh).transpose(1, 2)
        s = torch.matmul(q, k.transpose(2, 3)) * (1.0 / math.sqrt(dh)) + bias
        p = torch.softmax(s, dim=-1)
        ctx = r(torch.matmul(r(p), v).transpose(1, 2).reshape(M, L, d))
        s1 = lin(ctx, o_n) + x32
        if not mode["sum32"]:
            s1 = r(s1)
        x1_32 = F.layer_norm(s1, (d,), w[ln1_n + ".weight"], w[ln1_n + ".bias"], cfg.eps)
        x1b = r(x1_32)
        if not mode["res32"]:
            x1_32 = x1b
        h = r(F.gelu(lin(x1b, f1_n)))
        s2 = lin(h, f2_n) + x1_32
        if not mode["sum32"]:
            s2 = r(s2)
        x32 = F.layer_norm(s2, (d,), w[ln2_n + ".weight"], w[ln2_n + ".bias"], cfg.eps)
        cls = x32[:, 0, :]
        xb = r(x32)
        if not mode["res32"]:
            x32 = xb
    return cls


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
    bert = name == "cfg4"
    g = np.load(os.path.join(ROOT, "tests", "golden", "full_bert_cfg4.npz" if bert else f"full_distilbert_{name}.npz"))
    B, N, Lq, Lp = int(g["B"]), int(g["N"]), int(g["Lq"]), int(g["Lp"])
    cfg = E.RefConfig(arch="bert", n_layers=12) if bert else E.RefConfig()
    shapes = E.param_shapes(cfg)
    qp = {k: syn.init_param(11, k, s, std=0.02, perturb=True) for k, s in shapes.items()}
    pp = {k: syn.init_param(12, k, s, std=0.02, perturb=True) for k, s in shapes.items()}
    label_kind = str(g["label_kind"]) if "label_kind" in g else "teacher"
    batch = syn.nway_batch(4680, B, N, Lq, Lp, ragged=True, label_kind=label_kind)
    ref = g["logits"]
    amp = np.abs(g["logits_autocast_bf16"] - ref).max()
    print(f"{name}: max|logit| {np.abs(ref).max():.3f}; reference autocast drift {amp:.4f}")
    torch.set_num_threads(8)
    with torch.no_grad():
        variants = (("all bf16 (round 1)", dict(sum32=False, res32=False)), ("fp32 pre-LN sums", dict(sum32=True, res32=False)),
                    ("fp32 sums + fp32 residual", dict(sum32=True, res32=True)))
        if len(sys.argv) > 2:
            variants = variants[2:]
        for label, mode in variants:
            q = forward(qp, cfg, batch["query"]["input_ids"], batch["query"]["attention_mask"], mode)
            p = forward(pp, cfg, batch["nway_passages"]["input_ids"].reshape(B * N, Lp),
                        batch["nway_passages"]["attention_mask"].reshape(B * N, Lp), mode).view(B, N, -1)
            lg = torch.sum(q.unsqueeze(1) * p, dim=-1).numpy()
            err = np.abs(lg - ref).max()
            print(f"  {label:28s} max|dlogit| {err:.4f} = {err / amp:.2f} x reference autocast drift")


if __name__ == "__main__":
    main()
