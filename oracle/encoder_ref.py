"""Oracle (TEST INFRASTRUCTURE): pure-torch CPU restatement of the encoder + N-way scoring.

Follows:
  * ``models/nway_dual_encoder.py:21-67`` (forward / query_embs / passage_embs / nway_passage_embs,
    CLS pooling ``[0][:, 0, :]``, optional in-batch negatives ``:30-44``, dot-product scoring ``:47``);
  * the HuggingFace encoders the reference instantiates through ``AutoModel`` (third party, unpinned;
    installed transformers 5.15: ``models/distilbert/modeling_distilbert.py:82-282`` and
    ``models/bert/modeling_bert.py``): post-LN transformer, erf-GELU, additive key mask, LN eps 1e-12.

Parameters are plain dicts keyed by the HuggingFace state-dict names so reference checkpoints map 1:1.
Depends on torch + numpy only (no transformers import).  Differentiable through autograd, so the
same code is the gradient oracle.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class RefConfig:
    arch: str = "distilbert"          # "distilbert" | "bert"
    vocab_size: int = 30522
    dim: int = 768
    n_heads: int = 12
    hidden_dim: int = 3072
    n_layers: int = 6
    max_position_embeddings: int = 512
    type_vocab_size: int = 2
    eps: float = 1e-12


def param_shapes(cfg: RefConfig) -> dict:
    """HF state-dict names -> shapes (pooler omitted: it receives no gradient on this path)."""
    d, f = cfg.dim, cfg.hidden_dim
    s = {"embeddings.word_embeddings.weight": (cfg.vocab_size, d),
         "embeddings.position_embeddings.weight": (cfg.max_position_embeddings, d)}
    if cfg.arch == "bert":
        s["embeddings.token_type_embeddings.weight"] = (cfg.type_vocab_size, d)
    s["embeddings.LayerNorm.weight"] = (d,)
    s["embeddings.LayerNorm.bias"] = (d,)
    for i in range(cfg.n_layers):
        if cfg.arch == "distilbert":
            p = f"transformer.layer.{i}."
            names = [("attention.q_lin", (d, d)), ("attention.k_lin", (d, d)), ("attention.v_lin", (d, d)),
                     ("attention.out_lin", (d, d)), ("sa_layer_norm", None), ("ffn.lin1", (f, d)),
                     ("ffn.lin2", (d, f)), ("output_layer_norm", None)]
        else:
            p = f"encoder.layer.{i}."
            names = [("attention.self.query", (d, d)), ("attention.self.key", (d, d)),
                     ("attention.self.value", (d, d)), ("attention.output.dense", (d, d)),
                     ("attention.output.LayerNorm", None), ("intermediate.dense", (f, d)),
                     ("output.dense", (d, f)), ("output.LayerNorm", None)]
        for n, shp in names:
            if shp is None:
                s[p + n + ".weight"] = (d,)
                s[p + n + ".bias"] = (d,)
            else:
                s[p + n + ".weight"] = shp
                s[p + n + ".bias"] = (shp[0],)
    return s


def _layer_names(cfg: RefConfig, i: int):
    if cfg.arch == "distilbert":
        p = f"transformer.layer.{i}."
        return (p + "attention.q_lin", p + "attention.k_lin", p + "attention.v_lin", p + "attention.out_lin",
                p + "sa_layer_norm", p + "ffn.lin1", p + "ffn.lin2", p + "output_layer_norm")
    p = f"encoder.layer.{i}."
    return (p + "attention.self.query", p + "attention.self.key", p + "attention.self.value",
            p + "attention.output.dense", p + "attention.output.LayerNorm", p + "intermediate.dense",
            p + "output.dense", p + "output.LayerNorm")


def encoder_forward(params: dict, cfg: RefConfig, input_ids: torch.Tensor, attention_mask: torch.Tensor | None = None,
                    return_all: bool = False):
    """last_hidden_state [M, L, d] of the HF encoder in eval mode (dropout off), in the params' dtype."""
    M, L = input_ids.shape
    d, H = cfg.dim, cfg.n_heads
    dh = d // H
    w = params
    x = w["embeddings.word_embeddings.weight"][input_ids] + w["embeddings.position_embeddings.weight"][:L][None]
    if cfg.arch == "bert":
        x = x + w["embeddings.token_type_embeddings.weight"][0][None, None]
    x = F.layer_norm(x, (d,), w["embeddings.LayerNorm.weight"], w["embeddings.LayerNorm.bias"], cfg.eps)
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids)
    neg = torch.finfo(x.dtype).min
    bias = torch.zeros(M, 1, 1, L, dtype=x.dtype).masked_fill(attention_mask[:, None, None, :] == 0, neg)
    hiddens = [x]
    for i in range(cfg.n_layers):
        q_n, k_n, v_n, o_n, ln1_n, f1_n, f2_n, ln2_n = _layer_names(cfg, i)
        q = F.linear(x, w[q_n + ".weight"], w[q_n + ".bias"]).view(M, L, H, dh).transpose(1, 2)
        k = F.linear(x, w[k_n + ".weight"], w[k_n + ".bias"]).view(M, L, H, dh).transpose(1, 2)
        v = F.linear(x, w[v_n + ".weight"], w[v_n + ".bias"]).view(M, L, H, dh).transpose(1, 2)
        s = torch.matmul(q, k.transpose(2, 3)) * (1.0 / math.sqrt(dh)) + bias
        p = torch.softmax(s, dim=-1)
        ctx = torch.matmul(p, v).transpose(1, 2).reshape(M, L, d)
        sa = F.linear(ctx, w[o_n + ".weight"], w[o_n + ".bias"])
        x1 = F.layer_norm(sa + x, (d,), w[ln1_n + ".weight"], w[ln1_n + ".bias"], cfg.eps)
        h = F.gelu(F.linear(x1, w[f1_n + ".weight"], w[f1_n + ".bias"]))
        f = F.linear(h, w[f2_n + ".weight"], w[f2_n + ".bias"])
        x = F.layer_norm(f + x1, (d,), w[ln2_n + ".weight"], w[ln2_n + ".bias"], cfg.eps)
        hiddens.append(x)
    return (x, hiddens) if return_all else x


def cls_embs(params, cfg, enc) -> torch.Tensor:
    """``encoder(**enc)[0][:, 0, :]`` (reference models/nway_dual_encoder.py:51-57)."""
    return encoder_forward(params, cfg, enc["input_ids"], enc.get("attention_mask"))[:, 0, :]


def nway_passage_embs(params, cfg, nway_passages) -> torch.Tensor:
    """reference models/nway_dual_encoder.py:59-67."""
    ids, mask = nway_passages["input_ids"], nway_passages["attention_mask"]
    bz, nway, L = ids.shape
    reps = encoder_forward(params, cfg, ids.reshape(bz * nway, L), mask.reshape(bz * nway, L))[:, 0, :]
    return reps.view(bz, nway, -1)


def in_batch_index(bz: int, nway: int, all_in_batch_neg: bool) -> torch.Tensor:
    """Global passage index of every logit column (reference models/nway_dual_encoder.py:30-44).

    Returns int64 [bz, N'] with N' = bz*nway (all negatives) or 2*nway (next sample's passages, cyclic)."""
    own = torch.arange(bz)[:, None] * nway + torch.arange(nway)[None, :]
    if all_in_batch_neg:
        rows = []
        for b in range(bz):
            rows.append(torch.tensor(list(range(b * nway)) + list(range((b + 1) * nway, bz * nway)), dtype=torch.int64))
        neg = torch.stack(rows) if bz > 1 else torch.zeros(bz, 0, dtype=torch.int64)
    else:
        nxt = (torch.arange(bz) + 1) % bz
        neg = nxt[:, None] * nway + torch.arange(nway)[None, :]
    return torch.cat([own, neg], dim=1)


def nway_forward(q_params, p_params, cfg, queries, nway_passages, in_batch_loss=False, all_in_batch_neg=True):
    """logits [bz, N'] (reference models/nway_dual_encoder.py:21-49)."""
    q = cls_embs(q_params, cfg, queries)                       # [bz, D]
    P = nway_passage_embs(p_params, cfg, nway_passages)        # [bz, nway, D]
    if in_batch_loss:
        bz, nway, D = P.shape
        idx = in_batch_index(bz, nway, all_in_batch_neg)
        P = P.reshape(bz * nway, D)[idx]                       # [bz, N', D]
    return torch.sum(q.unsqueeze(1) * P, dim=-1)
