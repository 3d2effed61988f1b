"""``retriever.retrieval_utils`` of the reference (``retriever/retrieval_utils.py``) on MI355X.

Same call surface: ``get_embeddings_from_scratch`` (:30-58), ``construct_flatindex_from_embeddings`` (:116-129),
``index_retrieve`` (:131-153), ``convert_index_to_gpu`` (:155-184).  faiss is replaced by :class:`FlatIPIndex`
(``IndexIDMap(IndexFlatIP)`` semantics: exact fp32 inner product, results sorted by score descending, ids mapped,
missing results id -1) whose ``search`` runs on the GPU:

    1. threshold estimate  : bf16 MFMA scores of the queries against a strided row sample -> per-query k_s-th largest
    2. scan                : bf16 MFMA GEMM over the bf16 shadow of the whole shard with a filter epilogue that keeps
                             (row, score) pairs >= thr_q - eps_q  (HBM-bound: 2 B per index element per 128-query batch)
    3. exact re-score      : fp32 dot products of the candidates from the fp32 rows (what faiss would have computed)
    4. sort + cut          : (score desc, row position asc) -> top-k
    5. proof of exactness  : on the host, per query: thr_q <= (k-th exact candidate score) - eps_q with
                             eps_q = 2^-8 * |q| * max|p| >= |scan score - exact score|, so no row outside the candidate
                             list can be in the top-k; any query failing it (or overflowing the candidate buffer) is
                             rescanned with an adjusted threshold.

Multi-GPU: one process per GPU, rank r holds the contiguous row shard r; queries are replicated;
each rank returns its shard's top-k with GLOBAL ids and rank 0 merges on the host (score desc, tie -> lower id position).
The faiss sharding code of the reference (:164-182) is dead (undefined ``gpu_resources``); this is the design it intended.
"""
from __future__ import annotations

import math
import os
import pickle
from timeit import default_timer as timer

import numpy as np
import torch

from .. import hip_ops as ops

SAMPLE_ROWS = 65536          # rows scored for the threshold estimate
CAND_CAP = 8192              # candidate slots per query (LDS sort limit of cldrd_topk_sort)
QUERY_TILE = 128             # queries per scan (one MFMA tile row; the reference also searches in batches of 128)


def batch_to_device(batch, target_device: torch.device):
    for key in batch:
        if isinstance(batch[key], torch.Tensor):
            batch[key] = batch[key].to(target_device)
        if isinstance(batch[key], dict) or hasattr(batch[key], "keys"):
            for sub_key in batch[key]:
                if isinstance(batch[key][sub_key], torch.Tensor):
                    batch[key][sub_key] = batch[key][sub_key].to(target_device)
    return batch


def get_embeddings_from_scratch(model, dataloader, use_fp16, is_query, show_progress_bar=False):
    """Encode every batch of ``dataloader`` ({"seq": {input_ids, attention_mask}, "id": list[int]}) with the query or
    passage tower in eval mode -> (np.float32 [n, D], list[int]).  ``use_fp16`` is accepted for signature compatibility:
    the towers always run bf16 MFMA compute with fp32 CLS output (the reference's output is fp32 too, :56)."""
    embeddings, embeddings_ids = [], []
    model.eval()
    dev = next(model.parameters()).device
    pending = None
    for _, batch in enumerate(dataloader):
        with torch.no_grad():
            batch = batch_to_device(batch, dev)
            reps = model.query_embs(batch["seq"]) if is_query else model.passage_embs(batch["seq"])
            text_ids = batch["id"]
        # keep one batch in flight: the D2H copy of batch i overlaps the encode of batch i+1 (the reference syncs per batch, :47)
        host = torch.empty(reps.shape, dtype=torch.float32, pin_memory=True)
        host.copy_(reps, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        if pending is not None:
            pending[1].synchronize()
            embeddings.append(pending[0].numpy().copy())
        pending = (host, ev)
        assert isinstance(text_ids, list)
        embeddings_ids.extend(text_ids)
    if pending is not None:
        pending[1].synchronize()
        embeddings.append(pending[0].numpy().copy())
    embeddings = np.concatenate(embeddings)
    assert len(embeddings_ids) == embeddings.shape[0]
    assert isinstance(embeddings_ids[0], int)
    print(f"# nan in embeddings: {np.sum(np.isnan(embeddings))}")
    return embeddings, embeddings_ids


class FlatIPIndex:
    """Exact inner-product index over one shard of rows (faiss ``IndexIDMap(IndexFlatIP(d))`` semantics)."""

    def __init__(self, d: int):
        self.d = d
        self.ntotal = 0
        self.embeddings = None       # np.float32 [n, d] on the host until moved to a GPU
        self.ids = None              # np.int64 [n] or None (ids = row positions + id_offset)
        self.id_offset = 0
        self.device = None
        self._p32 = self._pbf = self._sample = None
        self.last_stats = {}
        self.profile = False          # bench.py: time the scan kernel with HIP events

    # -- construction -----------------------------------------------------------------------------------------
    def add_with_ids(self, embeddings, ids):
        emb = np.ascontiguousarray(embeddings, dtype=np.float32)
        if emb.ndim != 2 or emb.shape[1] != self.d:
            raise ValueError("embeddings must be [n, d]")
        ids = None if ids is None else np.asarray(ids).astype(np.int64)
        if ids is not None and ids.shape[0] != emb.shape[0]:
            raise ValueError("ids and embeddings differ in length")
        if self.embeddings is None:
            self.embeddings, self.ids = emb, ids
        else:
            self.embeddings = np.concatenate([self.embeddings, emb])
            self.ids = None if self.ids is None or ids is None else np.concatenate([self.ids, ids])
        self.ntotal = self.embeddings.shape[0]
        self._p32 = self._pbf = self._sample = None

    def add(self, embeddings):
        self.add_with_ids(embeddings, None)

    @classmethod
    def from_device_rows(cls, rows32: torch.Tensor, id_offset: int = 0) -> "FlatIPIndex":
        """Index over fp32 rows that already live in HBM (e.g. an encode shard that never left the GPU)."""
        idx = cls(rows32.shape[1])
        idx.ntotal, idx.id_offset = rows32.shape[0], id_offset
        idx._attach(rows32.contiguous())
        return idx

    def to_gpu(self, device):
        """Make the shard resident in HBM: fp32 rows (exact re-score), bf16 shadow (scan), bf16 row sample (threshold)."""
        device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("FlatIPIndex.search runs on the GPU only (no CPU path)")
        with torch.cuda.device(device):
            self._attach(torch.from_numpy(np.ascontiguousarray(self.embeddings)).to(device))
        return self

    def _attach(self, p32: torch.Tensor):
        device = p32.device
        self.device = device
        with torch.cuda.device(device):
            self._p32 = p32
            n, d = self._p32.shape
            self._pbf = torch.empty(n, d, dtype=torch.bfloat16, device=device)
            ops.cast_bf16(self._p32.view(-1), self._pbf.view(-1))
            self._s_stride = max(1, n // SAMPLE_ROWS)
            self._s_rows = min(n, (n + self._s_stride - 1) // self._s_stride)
            # the GEMM wants a column count that is a multiple of 8: zero rows pad the sample (never read by the select)
            self._sample = torch.zeros((self._s_rows + 7) // 8 * 8, d, dtype=torch.bfloat16, device=device)
            ops.gather_cast_rows(self._p32, self._sample, self._s_rows, self._s_stride)
            self._max_norm = math.sqrt(ops.row_sqnorm_max(self._p32))

    # -- search -------------------------------------------------------------------------------------------------
    def search(self, queries, k: int):
        """(D np.float32 [nq, k] descending, I np.int64 [nq, k]); missing results: id -1, score -inf."""
        if self._p32 is None:
            raise RuntimeError("index is not on a GPU: call convert_index_to_gpu(index, device) first (no CPU search path)")
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self.d:
            raise ValueError("queries must be [nq, d]")
        k = int(k)
        nq = q.shape[0]
        D = np.full((nq, k), -np.inf, dtype=np.float32)
        I = np.full((nq, k), -1, dtype=np.int64)
        stats = dict(scans=0, rescans=0, candidates=0)
        with torch.cuda.device(self.device):
            for lo in range(0, nq, QUERY_TILE):
                d_, rows = self._search_tile(q[lo:lo + QUERY_TILE], k, stats)
                kk = d_.shape[1]
                D[lo:lo + QUERY_TILE, :kk] = d_
                valid = rows >= 0
                if self.ids is None:
                    glob = np.where(valid, rows + self.id_offset, -1)
                else:
                    glob = np.where(valid, self.ids[np.maximum(rows, 0)], -1)
                I[lo:lo + QUERY_TILE, :kk] = glob
        if "scan_events" in stats:
            torch.cuda.synchronize()
            stats["scan_ms"] = [a.elapsed_time(b) for a, b in stats.pop("scan_events")]
        self.last_stats = stats
        return D, I

    def _search_tile(self, q_np: np.ndarray, k: int, stats):
        dev = self.device
        nq, d = q_np.shape
        n = self._p32.shape[0]
        kk = min(k, n, CAND_CAP)
        q32 = torch.from_numpy(q_np).to(dev)
        qb = torch.empty(nq, d, dtype=torch.bfloat16, device=dev)
        ops.cast_bf16(q32.view(-1), qb.view(-1))
        eps = (2.0 ** -8) * np.linalg.norm(q_np.astype(np.float64), axis=1) * self._max_norm + 1e-30
        # 1. threshold estimate from the row sample
        S = self._s_rows
        samp = torch.empty(nq, self._sample.shape[0], dtype=torch.float32, device=dev)
        ops.gemm_nt(qb, self._sample, samp, nq)
        lam = kk * S / n
        # lam = expected number of sample rows inside the global top-kk; +4.5 sigma makes a too-high estimate (-> rescan)
        # a ~1e-5 event per query at the price of ~25 % more candidates
        kth = int(min(S, math.ceil(lam + 4.5 * math.sqrt(lam) + 1.0))) if S < n else kk
        thr_dev = torch.empty(nq, dtype=torch.float32, device=dev)
        ops.topk_kth_largest(samp, S, kth, thr_dev)
        thr = thr_dev.cpu().numpy().astype(np.float64) - eps
        counts = torch.empty(nq + 1, dtype=torch.int32, device=dev)      # [nq]: hits the streaming scan had to drop
        cand_rows = torch.empty(nq, CAND_CAP, dtype=torch.int32, device=dev)
        cand_scores = torch.empty(nq, CAND_CAP, dtype=torch.float32, device=dev)
        D = torch.empty(nq, kk, dtype=torch.float32, device=dev)
        I = torch.empty(nq, kk, dtype=torch.int32, device=dev)
        out_D = np.empty((nq, kk), dtype=np.float32)
        out_I = np.empty((nq, kk), dtype=np.int64)
        todo = np.ones(nq, dtype=bool)
        for attempt in range(12):
            thr_dev.copy_(torch.from_numpy(thr.astype(np.float32)))
            counts.zero_()
            if self.profile:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            ops.topk_scan_filter(qb, self._pbf, thr_dev, counts, cand_rows, cand_scores)      # 2. scan
            if self.profile:
                e1.record()
                stats.setdefault("scan_events", []).append((e0, e1))
            ops.topk_rescore(q32, self._p32, counts, cand_rows, cand_scores)                    # 3. exact re-score
            ops.topk_sort(counts, cand_rows, cand_scores, kk, D, I)                             # 4. sort + cut
            stats["scans"] += 1
            stats["rescans"] += attempt > 0
            c = counts.cpu().numpy()
            if c[nq] != 0:        # the streaming kernel's on-chip hit list overflowed: redo this scan with the tiled kernel
                counts.zero_()
                ops.topk_scan_filter(qb, self._pbf, thr_dev, counts, cand_rows, cand_scores, tiled=True)
                ops.topk_rescore(q32, self._p32, counts, cand_rows, cand_scores)
                ops.topk_sort(counts, cand_rows, cand_scores, kk, D, I)
                stats["scans"] += 1
                stats["tiled_rescans"] = stats.get("tiled_rescans", 0) + 1
                c = counts.cpu().numpy()
            c = c[:nq]
            Dh, Ih = D.cpu().numpy(), I.cpu().numpy().astype(np.int64)
            stats["candidates"] += int(np.minimum(c, CAND_CAP)[todo].sum())
            # 5. proof of exactness per query
            kth_exact = Dh[:, kk - 1].astype(np.float64)
            enough = c >= kk
            overflow = c > CAND_CAP
            proven = enough & ~overflow & (thr <= kth_exact - eps)
            done_now = todo & proven
            out_D[done_now], out_I[done_now] = Dh[done_now], Ih[done_now]
            todo &= ~proven
            if not todo.any():
                break
            # adjust: too few / not proven -> lower the threshold below the proven bound; overflow -> raise it half-way
            low = todo & ~overflow
            thr[low] = np.where(enough[low], kth_exact[low] - 2.0 * eps[low], thr[low] - np.maximum(4.0 * eps[low], 0.05 * np.abs(thr[low]) + 1e-3))
            ov = todo & overflow
            thr[ov] = thr[ov] + 0.5 * eps[ov] + 1e-3 * np.abs(thr[ov])
            if attempt >= 8:
                thr[todo & ~overflow] = -np.inf      # exhaustive: every row is a candidate (only valid for n <= CAND_CAP)
        else:
            raise RuntimeError("top-k search did not converge (candidate buffer too small for this score distribution)")
        return out_D, out_I

    # -- persistence (own format; faiss' binary layout is not reproduced) ----------------
    def write(self, path: str):
        np.save(path + ".emb.npy", self.embeddings)
        with open(path + ".meta.pkl", "wb") as fh:
            pickle.dump({"d": self.d, "ids": self.ids, "id_offset": self.id_offset, "format": "cldrd-flatip-v1"}, fh)

    @classmethod
    def read(cls, path: str) -> "FlatIPIndex":
        with open(path + ".meta.pkl", "rb") as fh:
            meta = pickle.load(fh)
        idx = cls(meta["d"])
        idx.embeddings = np.load(path + ".emb.npy", mmap_mode="r")
        idx.ids, idx.id_offset = meta["ids"], meta.get("id_offset", 0)
        idx.ntotal = idx.embeddings.shape[0]
        return idx


def write_index(index: FlatIPIndex, path: str, faiss_format: bool = False):
    """``faiss.write_index`` of the reference (index_text.py:103).  Default: this package's memory-mappable pair of files;
    ``faiss_format=True`` writes faiss' own ``IndexIDMap(IndexFlatIP)`` serialisation to ``path`` (see ``write_faiss_index``)."""
    if faiss_format:
        write_faiss_index(index, path)
    else:
        index.write(path)


def read_index(path: str) -> FlatIPIndex:
    """``faiss.read_index`` of the reference (retrieve_top_passages.py:77): reads either format (a faiss file starts with the
    fourcc ``IxMp`` / ``IxM2`` / ``IxFI``)."""
    if os.path.isfile(path):
        with open(path, "rb") as fh:
            magic = fh.read(4)
        if magic in (b"IxMp", b"IxM2", b"IxFI"):
            return read_faiss_index(path)
    return FlatIPIndex.read(path)


# ---- faiss binary interop -----------------------------------------------------------------
# Layout restated from faiss' published serialisation (faiss/impl/index_write.cpp, v1.7+), little endian:
#   index header : int32 d | int64 ntotal | int64 dummy (1 << 20) | int64 dummy | uint8 is_trained | int32 metric_type
#                  (0 = inner product, 1 = L2; a float32 metric_arg follows only for metric_type > 1)
#   "IxFI" flat  : header | uint64 n_floats (= ntotal * d) | n_floats float32, row major
#   "IxMp" idmap : header | <nested index> | uint64 n_ids | n_ids int64            ("IxM2" = IndexIDMap2, same payload)
# PARITY UNPINNED: faiss is not installed in the build image, so these two functions are checked against each other and
# against a hand-assembled byte string only (tests/test_oracle_optim_retrieval.py), not against a file written by faiss.
def _faiss_header(d: int, ntotal: int) -> bytes:
    import struct
    return struct.pack("<iqqqBi", d, ntotal, 1 << 20, 1 << 20, 1, 0)


def write_faiss_index(index: "FlatIPIndex", path: str):
    import struct
    emb = np.ascontiguousarray(index.embeddings, dtype=np.float32)
    n, d = emb.shape
    ids = np.asarray(index.ids if index.ids is not None else np.arange(n) + index.id_offset, dtype=np.int64)
    with open(path, "wb") as fh:
        fh.write(b"IxMp" + _faiss_header(d, n))
        fh.write(b"IxFI" + _faiss_header(d, n) + struct.pack("<Q", n * d))
        fh.write(emb.tobytes())
        fh.write(struct.pack("<Q", n))
        fh.write(ids.tobytes())


def read_faiss_index(path: str) -> "FlatIPIndex":
    import struct

    def header(fh):
        d, ntotal, _, _, trained, metric = struct.unpack("<iqqqBi", fh.read(33))
        if metric > 1:
            fh.read(4)
        if metric != 0:
            raise ValueError("only inner-product flat indexes are supported (the reference builds IndexFlatIP)")
        return d, ntotal

    def flat(fh):
        d, ntotal = header(fh)
        (nfl,) = struct.unpack("<Q", fh.read(8))
        if nfl != ntotal * d:
            raise ValueError("faiss flat index: vector size does not match ntotal * d")
        return d, np.frombuffer(fh.read(4 * nfl), dtype=np.float32).reshape(ntotal, d)

    with open(path, "rb") as fh:
        magic = fh.read(4)
        if magic in (b"IxMp", b"IxM2"):
            header(fh)
            if fh.read(4) != b"IxFI":
                raise ValueError("faiss id map: nested index is not IndexFlatIP")
            d, emb = flat(fh)
            (nid,) = struct.unpack("<Q", fh.read(8))
            ids = np.frombuffer(fh.read(8 * nid), dtype=np.int64)
        elif magic == b"IxFI":
            d, emb = flat(fh)
            ids = None
        else:
            raise ValueError(f"not a faiss flat inner-product index (fourcc {magic!r})")
    idx = FlatIPIndex(d)
    if ids is None:
        idx.add(emb)
    else:
        idx.add_with_ids(emb, ids)
    return idx


def construct_flatindex_from_embeddings(embeddings, ids):
    """reference :116-129: flat inner-product index (+ id map when ``ids`` is given)."""
    hidden_size = embeddings.shape[1]
    print("embedding shape: " + str(embeddings.shape))
    index = FlatIPIndex(hidden_size)
    if ids is not None:
        if isinstance(ids, list):
            ids = np.array(ids)
        ids = ids.astype(np.int64)
        print(ids.shape, ids.dtype)
        index.add_with_ids(embeddings, ids)
    else:
        index.add(embeddings)
    return index


class ShardedFlatIPIndex:
    """Row-sharded index: this process holds rows [lo, hi) of the global matrix on its GPU; ``search`` returns the global
    top-k on rank 0 (other ranks get their local lists).  Works without torch.distributed as a single shard."""

    def __init__(self, local: FlatIPIndex, rank: int = 0, world: int = 1, group=None):
        self.local, self.rank, self.world, self.group = local, rank, world, group
        self.ntotal = local.ntotal

    @staticmethod
    def shard_bounds(n: int, world: int, rank: int):
        per = -(-n // world)
        return min(n, rank * per), min(n, (rank + 1) * per)

    def search(self, queries, k):
        D, I = self.local.search(queries, k)
        if self.world == 1:
            return D, I
        import torch.distributed as dist
        gathered = [None] * self.world if self.rank == 0 else None
        dist.gather_object((D, I), gathered, dst=0, group=self.group)
        if self.rank != 0:
            return D, I
        return merge_shard_results([g[0] for g in gathered], [g[1] for g in gathered], k)


def merge_shard_results(shard_D, shard_I, k):
    """Host k-way merge of per-shard top-k lists: score desc, tie -> lower global id; missing entries (id -1) last."""
    D = np.concatenate(shard_D, axis=1).astype(np.float64)
    I = np.concatenate(shard_I, axis=1)
    key_i = np.where(I < 0, np.iinfo(np.int64).max, I)
    order = np.lexsort((key_i, -D), axis=1)[:, :k]
    return np.take_along_axis(D, order, axis=1).astype(np.float32), np.take_along_axis(I, order, axis=1)


def convert_index_to_gpu(index, faiss_gpu_index, useFloat16=False):
    """reference :155-184.  int (or 1-element list): whole index on that GPU.  list of several devices: not supported in one
    process - shard with one process per GPU and :class:`ShardedFlatIPIndex` instead.  ``useFloat16`` is ignored: the scan
    already reads a 16-bit shadow and the returned scores are exact fp32 either way."""
    if type(faiss_gpu_index) == list and len(faiss_gpu_index) == 1:
        faiss_gpu_index = faiss_gpu_index[0]
    if isinstance(faiss_gpu_index, int):
        return index.to_gpu(faiss_gpu_index)
    raise NotImplementedError("multi-GPU search is one process per GPU: see ShardedFlatIPIndex / retrieve_top_passages.py")


def index_retrieve(index, query_embeddings, topk, batch=None):
    """reference :131-153: search everything at once or in query batches; returns (scores, ids) as nested lists when
    batched (as the reference does), arrays otherwise."""
    print("Query Num", len(query_embeddings))
    start = timer()
    if batch is None:
        nn_scores, nearest_neighbors = index.search(query_embeddings, topk)
    else:
        query_offset_base = 0
        nearest_neighbors = []
        nn_scores = []
        while query_offset_base < len(query_embeddings):
            batch_query_embeddings = query_embeddings[query_offset_base:query_offset_base + batch]
            batch_nn_scores, batch_nn = index.search(batch_query_embeddings, topk)
            nearest_neighbors.extend(batch_nn.tolist())
            nn_scores.extend(batch_nn_scores.tolist())
            query_offset_base += len(batch_query_embeddings)
    elapsed_time = timer() - start
    elapsed_time_per_query = 1000 * elapsed_time / len(query_embeddings)
    print(f"Elapsed Time: {elapsed_time:.1f}s, Elapsed Time per query: {elapsed_time_per_query:.1f}ms")
    return nn_scores, nearest_neighbors
