"""Oracle (TEST INFRASTRUCTURE): exact inner-product top-k and the batched search loop.

Follows ``retriever/retrieval_utils.py:116-153`` (flat inner-product index with an id map, searched in
query batches) and the run-file layout of ``retriever/retrieve_top_passages.py:90-107``.

PARITY UNPINNED: the arithmetic lives in faiss (``IndexFlatIP`` + ``IndexIDMap``; third party, not
installed here, no pinned version).  Restated contract: scores = exact fp32 inner products, results
per query sorted by score descending, missing results padded with id -1 / score -inf.  Tie order in
faiss is implementation defined; this oracle (and the build) fix "score desc, then row position asc".
Scores are accumulated in float64 and rounded to float32 once.
"""
from __future__ import annotations

import numpy as np


def flat_ip_search(embeddings: np.ndarray, ids: np.ndarray | None, queries: np.ndarray, k: int,
                   chunk: int = 65536):
    """Return (D float32 [nq, k] descending, I int64 [nq, k]) for the exact inner product."""
    emb = np.asarray(embeddings)
    q = np.asarray(queries, dtype=np.float64)
    n = emb.shape[0]
    nq = q.shape[0]
    best_s = np.full((nq, 0), -np.inf)
    best_i = np.zeros((nq, 0), dtype=np.int64)
    for lo in range(0, n, chunk):
        s = (q @ emb[lo:lo + chunk].astype(np.float64).T).astype(np.float32).astype(np.float64)
        idx = np.broadcast_to(np.arange(lo, lo + s.shape[1], dtype=np.int64), s.shape)
        cs = np.concatenate([best_s, s], axis=1)
        ci = np.concatenate([best_i, idx], axis=1)
        order = np.lexsort((ci, -cs), axis=1)[:, :k]          # score desc, then position asc
        best_s = np.take_along_axis(cs, order, axis=1)
        best_i = np.take_along_axis(ci, order, axis=1)
    D = np.full((nq, k), -np.inf, dtype=np.float32)
    I = np.full((nq, k), -1, dtype=np.int64)
    kk = best_s.shape[1]
    D[:, :kk] = best_s.astype(np.float32)
    I[:, :kk] = best_i if ids is None else np.asarray(ids, dtype=np.int64)[best_i]
    return D, I


class FlatIPIndex:
    """Minimal stand-in for ``faiss.IndexIDMap(faiss.IndexFlatIP(d))`` (``search`` only)."""

    def __init__(self, embeddings, ids=None):
        self.embeddings = np.ascontiguousarray(embeddings, dtype=np.float32)
        self.ids = None if ids is None else np.asarray(ids, dtype=np.int64)
        self.ntotal = self.embeddings.shape[0]

    def search(self, queries, k):
        return flat_ip_search(self.embeddings, self.ids, queries, k)


def index_retrieve(index, query_embeddings, topk, batch=None):
    """Batched search loop returning Python lists (reference retriever/retrieval_utils.py:131-153)."""
    if batch is None:
        D, I = index.search(query_embeddings, topk)
        return D, I
    nn_scores, nearest = [], []
    base = 0
    while base < len(query_embeddings):
        qb = query_embeddings[base:base + batch]
        D, I = index.search(qb, topk)
        nearest.extend(I.tolist())
        nn_scores.extend(D.tolist())
        base += len(qb)
    return nn_scores, nearest


def run_file_lines(query_ids, nn_doc_ids, nn_scores):
    """``qid\\tdocid\\trank\\tscore`` lines, rank 1-based (reference retriever/retrieve_top_passages.py:90-107)."""
    lines = []
    for qid, docids, scores in zip(query_ids, nn_doc_ids, nn_scores):
        for i, (docid, s) in enumerate(zip(docids, scores)):
            lines.append(f"{qid}\t{docid}\t{i + 1}\t{s}\n")
    return lines


def merge_shard_results(shard_D, shard_I, k):
    """Host merge of per-shard top-k lists: score desc; ties -> shard asc, then position in the shard's list asc (for contiguous row-range
    shards whose lists are ordered (score desc, row asc) that IS global row position asc, the tie rule of flat_ip_search above; with
    ids = row positions - every test, and the CLIs - it is also "lower id first", the rule this function had until round 4); missing
    entries (id -1) last; (-inf, -1) padding when fewer than k candidates exist."""
    D = np.concatenate(shard_D, axis=1).astype(np.float64)
    I = np.concatenate(shard_I, axis=1)
    pos = np.broadcast_to(np.arange(D.shape[1], dtype=np.int64), D.shape)
    missing = (I < 0)
    order = np.lexsort((pos, -D, missing), axis=1)[:, :k]
    Dm = np.take_along_axis(D, order, axis=1).astype(np.float32)
    Im = np.take_along_axis(I, order, axis=1)
    Dm[Im < 0] = -np.inf
    if Dm.shape[1] < k:
        pad = k - Dm.shape[1]
        Dm = np.concatenate([Dm, np.full((Dm.shape[0], pad), -np.inf, dtype=np.float32)], axis=1)
        Im = np.concatenate([Im, np.full((Im.shape[0], pad), -1, dtype=np.int64)], axis=1)
    return Dm, Im
