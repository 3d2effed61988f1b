"""Portable synthetic data / weight generator for the hot path's input contract.

Replaces ``NwayDataset.collate_fn`` (reference ``dataset/nway_dataset.py:87-118``) and the label
schemes (``dataset/nway_dataset.py:41-72``) with tensors of the same layout, produced by a
counter-based generator (splitmix64 -> uniform -> Box-Muller, all in integer/float64 numpy) so the
same seeds give the same tensors on any torch build.
"""
from __future__ import annotations

import numpy as np
import torch

CLS_ID = 101
SEP_ID = 102
BODY_LO = 1000
VOCAB = 30522

_U64 = np.uint64
_MASK = _U64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """n outputs of splitmix64 whose state starts at ``seed`` (stateless / counter based)."""
    with np.errstate(over="ignore"):
        idx = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = (_U64(seed & 0xFFFFFFFFFFFFFFFF) + idx * _U64(0x9E3779B97F4A7C15)) & _MASK
        z = ((z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)) & _MASK
        z = z ^ (z >> _U64(31))
    return z


def uniform01(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """float64 uniforms in [0, 1) with 53 random bits."""
    return (splitmix64(seed, n, offset) >> _U64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """float64 standard normals (Box-Muller on pairs of uniforms)."""
    m = (n + 1) // 2
    u = uniform01(seed, 2 * m, 2 * offset)
    u1 = 1.0 - u[0::2]          # (0, 1]
    u2 = u[1::2]
    r = np.sqrt(-2.0 * np.log(u1))
    out = np.empty(2 * m, dtype=np.float64)
    out[0::2] = r * np.cos(2.0 * np.pi * u2)
    out[1::2] = r * np.sin(2.0 * np.pi * u2)
    return out[:n]


def randint(seed: int, lo: int, hi: int, n: int, offset: int = 0) -> np.ndarray:
    """int64 uniform integers in [lo, hi)."""
    return (lo + (splitmix64(seed, n, offset) % _U64(hi - lo)).astype(np.int64)).astype(np.int64)


def _name_seed(seed: int, name: str) -> int:
    h = 1469598103934665603
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return (h ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF


def init_tensor(seed: int, name: str, shape, std: float = 0.02) -> torch.Tensor:
    """fp32 N(0, std) tensor keyed by (seed, parameter name)."""
    n = int(np.prod(shape))
    return torch.from_numpy((normal(_name_seed(seed, name), n) * std).astype(np.float32).reshape(shape))


def token_ids(seed: int, rows: int, length: int, vocab: int = VOCAB, body_lo: int = BODY_LO) -> np.ndarray:
    """[rows, length] int64: [CLS] body... [SEP]."""
    lo = body_lo if body_lo < vocab - 1 else 3
    ids = randint(seed, lo, vocab, rows * length).reshape(rows, length)
    ids[:, 0] = CLS_ID if vocab > CLS_ID else 1
    ids[:, -1] = SEP_ID if vocab > SEP_ID else 2
    return ids


def msmarco_lengths(seed: int, rows: int, max_len: int) -> np.ndarray:
    """True token counts ~ clip(round(LogNormal(4.3, 0.35)), 16, max_len) (median ~74)."""
    z = normal(seed, rows)
    return np.clip(np.rint(np.exp(4.3 + 0.35 * z)), min(16, max_len), max_len).astype(np.int64)


def labels_mode9(batch: int, nway: int) -> np.ndarray:
    """Generalised label mode 9 (reference dataset/nway_dataset.py:65-67): first ceil(N/3) = 1/rank,
    next floor(N/3) = -0.25, rest -0.5."""
    n_rel = -(-nway // 3)
    n_mid = nway // 3
    row = np.concatenate([1.0 / np.arange(1, n_rel + 1), np.full(n_mid, -0.25), np.full(nway - n_rel - n_mid, -0.5)])
    return np.tile(row.astype(np.float32), (batch, 1))


def teacher_scores(seed: int, batch: int, nway: int) -> np.ndarray:
    """Teacher scores for kl_div / margin_mse: sort_desc(N(0,1)*4 + 8) per row."""
    t = normal(seed, batch * nway).reshape(batch, nway) * 4.0 + 8.0
    return (-np.sort(-t, axis=1)).astype(np.float32)


def nway_batch(seed: int, batch: int, nway: int, q_len: int, p_len: int, *, vocab: int = VOCAB,
               ragged: bool = False, label_kind: str = "teacher") -> dict:
    """A training batch with the collate_fn layout (reference dataset/nway_dataset.py:103-118)."""
    q_ids = token_ids(seed + 1, batch, q_len, vocab)
    p_ids = token_ids(seed + 2, batch * nway, p_len, vocab)
    q_mask = np.ones_like(q_ids)
    p_mask = np.ones_like(p_ids)
    if ragged:
        lens = msmarco_lengths(seed + 3, batch * nway, p_len)
        ar = np.arange(p_len)[None, :]
        p_mask = (ar < lens[:, None]).astype(np.int64)
        sep_pos = lens - 1
        p_ids = np.where(ar < lens[:, None], p_ids, 0)
        p_ids[np.arange(batch * nway), sep_pos] = SEP_ID if vocab > SEP_ID else 2
        qlens = np.clip(msmarco_lengths(seed + 4, batch, q_len) // 8, 4, q_len)
        arq = np.arange(q_len)[None, :]
        q_mask = (arq < qlens[:, None]).astype(np.int64)
        q_ids = np.where(arq < qlens[:, None], q_ids, 0)
        q_ids[np.arange(batch), qlens - 1] = SEP_ID if vocab > SEP_ID else 2
    labels = teacher_scores(seed + 5, batch, nway) if label_kind == "teacher" else labels_mode9(batch, nway)
    return {
        "qid": torch.arange(batch, dtype=torch.int64),
        "query": {"input_ids": torch.from_numpy(q_ids), "attention_mask": torch.from_numpy(q_mask)},
        "nway_passages": {"input_ids": torch.from_numpy(p_ids.reshape(batch, nway, p_len)),
                          "attention_mask": torch.from_numpy(p_mask.reshape(batch, nway, p_len))},
        "labels": torch.from_numpy(labels),
    }


def seq_batch(seed: int, rows: int, length: int, *, vocab: int = VOCAB, ragged: bool = False, first_id: int = 0) -> dict:
    """An encode batch with the SequenceDataset.collate_fn layout (reference dataset/sequence_dataset.py:44-55)."""
    ids = token_ids(seed, rows, length, vocab)
    mask = np.ones_like(ids)
    if ragged:
        lens = msmarco_lengths(seed + 1, rows, length)
        ar = np.arange(length)[None, :]
        mask = (ar < lens[:, None]).astype(np.int64)
        ids = np.where(ar < lens[:, None], ids, 0)
        ids[np.arange(rows), lens - 1] = SEP_ID if vocab > SEP_ID else 2
    return {"seq": {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)},
            "id": list(range(first_id, first_id + rows))}


def seq_rows(seed: int, first_row: int, rows: int, length: int, *, vocab: int = VOCAB, ragged: bool = False) -> dict:
    """Rows [first_row, first_row + rows) of ONE endless synthetic collection: row r is a function of (seed, r) only, so any
    sharding / batching of the collection encodes the same passages (index_text.py under RANK / WORLD_SIZE must give the rows the
    single-process run gives).  Same layout as :func:`seq_batch`, ids = global row numbers."""
    lo = BODY_LO if BODY_LO < vocab - 1 else 3
    ids = randint(seed, lo, vocab, rows * length, offset=first_row * length).reshape(rows, length)
    ids[:, 0] = CLS_ID if vocab > CLS_ID else 1
    ids[:, -1] = SEP_ID if vocab > SEP_ID else 2
    mask = np.ones_like(ids)
    if ragged:
        z = normal(seed + 1, 2 * rows, offset=first_row)[0::2]          # Box-Muller pair (first_row + j) -> row first_row + j
        lens = np.clip(np.rint(np.exp(4.3 + 0.35 * z)), min(16, length), length).astype(np.int64)
        ar = np.arange(length)[None, :]
        mask = (ar < lens[:, None]).astype(np.int64)
        ids = np.where(ar < lens[:, None], ids, 0)
        ids[np.arange(rows), lens - 1] = SEP_ID if vocab > SEP_ID else 2
    return {"seq": {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)},
            "id": list(range(first_row, first_row + rows))}


def corpus_embeddings(seed: int, rows: int, dim: int = 768, offset_rows: int = 0) -> np.ndarray:
    """CLS-like rows: unit-variance Gaussian direction scaled to a per-row norm ~ U(9, 12)."""
    g = normal(seed, rows * dim, offset_rows * dim).reshape(rows, dim)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    norms = 9.0 + 3.0 * uniform01(seed ^ 0x5DEECE66D, rows, offset_rows)
    return (g * norms[:, None]).astype(np.float32)


def init_param(seed: int, name: str, shape, std: float = 0.02, perturb: bool = True) -> torch.Tensor:
    """Seeded test initialisation keyed by HF parameter name.  ``perturb=True`` (parity tests) gives LayerNorm
    gains 1 + N(0, 0.1) and biases N(0, 0.05) so no term of the backward is trivially zero; ``perturb=False``
    is the HF init (LN gain 1, all biases 0, weights N(0, std))."""
    is_ln_w = "LayerNorm.weight" in name or "layer_norm.weight" in name
    is_bias = name.endswith(".bias")
    if perturb:
        if is_ln_w:
            return 1.0 + init_tensor(seed, name, shape, std=0.1)
        if is_bias:
            return init_tensor(seed, name, shape, std=0.05)
        return init_tensor(seed, name, shape, std=std)
    if is_ln_w:
        return torch.ones(shape, dtype=torch.float32)
    if is_bias:
        return torch.zeros(shape, dtype=torch.float32)
    return init_tensor(seed, name, shape, std=std)


def cls_like_corpus(rows: int, d: int, seed: int, device, chunk: int = 1 << 20):
    """CLS-like (anisotropic) embeddings on `device`, torch's generator (timing / regime workloads, not golden data): a dominant common
    direction u scaled by 3.6 U(0.8, 1.2) per row plus 0.045 N(0, I), tuned so that one query's scores over the corpus have std / mean
    ~ 0.12 - what the reference model's own CLS vectors show (tests/golden/full_distilbert_cfg2.npz: q_cls . p_cls = 17 +- 2 per row).
    Every row scores close to every other: the regime an isotropic corpus (k-th score far out in a thin tail) does not exercise.
    Returns (P fp32 [rows, d], u [d])."""
    gen = torch.Generator(device=device).manual_seed(seed)
    u = torch.randn(d, device=device, generator=gen)
    u /= u.norm()
    P = torch.empty(rows, d, device=device)
    for lo in range(0, rows, chunk):
        m = min(chunk, rows - lo)
        a = 3.6 * (0.8 + 0.4 * torch.rand(m, 1, device=device, generator=gen))
        P[lo:lo + m] = a * u + 0.045 * torch.randn(m, d, device=device, generator=gen)
    return P, u


def cls_like_queries(nq: int, u: torch.Tensor, seed: int):
    """Queries for `cls_like_corpus`: 4.6 U(0.9, 1.1) u + 0.06 N(0, I)."""
    gen = torch.Generator(device=u.device).manual_seed(seed)
    return 4.6 * (0.9 + 0.2 * torch.rand(nq, 1, device=u.device, generator=gen)) * u + 0.06 * torch.randn(nq, u.shape[0], device=u.device, generator=gen)

