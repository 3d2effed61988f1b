"""N-way training examples -> batches with the contract of reference ``dataset/nway_dataset.py``.

Same class / constructor / key names as the reference so ``trainer/nway_listwise`` reads like ``nway_listwise_1.py:173-245``:

* ``NwayDataset.__getitem__`` (reference :32-83): one query, its ``relT_pids`` then ``neg_pids`` passages, and the label row of the
  ``label_mode`` (table below == reference :41-72);
* ``collate_fn`` (reference :88-118): tokenises queries to ``max_query_len`` and the flattened passages to ``max_passage_len``
  (padding to the longest in the batch, truncation 'longest_first'), views passages as ``[B, N, L]``;
* the file constructors (reference :120-470) differ only in the JSON field that holds the negatives and in which label modes
  they accept; all distributed variants keep line ``i`` on rank ``i % nranks`` (reference :305).

Host tokenisation becomes the bottleneck at MI355X speeds, so the same batches can also come
from ``TokenCache``: token ids of every query / passage tokenised ONCE into memory-mapped ``[n, L]`` int32 arrays plus lengths;
``NwayDataset(..., token_cache=...)`` then builds batches by gather + trim, with output identical to the tokenising path
(tests/test_nway_dataset.py).
"""
from __future__ import annotations

import json
import os
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

# label_mode -> (number of relT passages, number of negatives, labels of the negatives); relT labels are listed per mode below
_NEG = {
    "1": [0.0] * 5,
    "2": [0.5] * 10 + [0.0] * 10,
    "3": [0.0] * 20,
    "4": [0.5] * 10 + [0.0] * 10,
    "5": [0.0] * 10,
    "6": [],
    "7": [0.0] * 25,
    "8": [-0.25] * 12 + [-0.5] * 13,
    "9": [-0.25] * 10 + [-0.5] * 10,
    "10": [-0.25] * 5 + [-0.5] * 5,
}
_REL = {"1": 1, "2": 10, "3": 10, "4": 10, "5": 20, "6": 30, "7": 5, "8": 5, "9": 10, "10": 20}
LABEL_MODES = tuple(_REL)


def labels_for_mode(label_mode: str) -> List[float]:
    """The label row of one example (reference nway_dataset.py:41-72): reciprocal rank over the relT passages except modes 1, 2
    (all ones) and 4 (1, then 0.9); negatives per the table above."""
    if label_mode not in _REL:
        raise ValueError(f"{label_mode} do not defined")
    n_rel = _REL[label_mode]
    if label_mode in ("1", "2"):
        rel = [1.0] * n_rel
    elif label_mode == "4":
        rel = [1.0] + [0.9] * (n_rel - 1)
    else:
        rel = list(1.0 / np.arange(1, 1 + n_rel))
    return rel + list(_NEG[label_mode])


def _read_queries(path: str) -> Dict[int, str]:
    out = {}
    with open(path, "r") as fh:
        for line in fh:
            parts = line.strip().split("\t")
            out[int(parts[0])] = parts[1]
    return out


def _read_passages(path: str) -> Dict[int, object]:
    """``pid \\t passage`` or ``pid \\t title \\t para`` (reference :129-139; the 3-column form yields a dict)."""
    out = {}
    with open(path, "r") as fh:
        for line in fh:
            parts = line.strip().split("\t")
            if len(parts) == 2:
                out[int(parts[0])] = parts[1]
            elif len(parts) == 3:
                out[int(parts[0])] = {"title": parts[1], "para": parts[2]}
            else:
                raise ValueError("array {}, with illegal length".format(parts))
    return out


def _check_rank(rank: int, nranks: Optional[int]):
    if rank != -1 and not (nranks is not None and nranks > 1 and 0 <= rank < nranks):
        raise AssertionError("rank must be in range(nranks) with nranks > 1")


def _read_json_lines(path: str, rank: int, nranks: Optional[int], convert: Callable[[dict], dict]) -> List[dict]:
    out = []
    with open(path, "r") as fh:
        for i, line in enumerate(fh):
            if rank == -1 or i % nranks == rank:
                out.append(convert(json.loads(line)))
    return out


def _hard_negatives(ex: dict) -> dict:
    return {"qid": ex["qid"], "relT_pids": ex["relT_pids"], "neg_pids": ex["most_hard_pids"] + ex["semi_hard_pids"]}


class TokenCache:
    """Token ids of a text table tokenised once: ``ids[n, L]`` int32 (zero padded), ``lens[n]`` int32, ``keys[n]`` int64 (qid / pid).
    Stored as three ``.npy`` files next to each other and memory-mapped on load."""

    def __init__(self, keys: np.ndarray, ids: np.ndarray, lens: np.ndarray):
        self.keys, self.ids, self.lens = keys, ids, lens
        self._row = {int(k): i for i, k in enumerate(keys)}

    @classmethod
    def build(cls, table: Dict[int, object], tokenizer, max_len: int, chunk: int = 4096) -> "TokenCache":
        keys = np.fromiter(table.keys(), dtype=np.int64, count=len(table))
        ids = np.zeros((len(keys), max_len), dtype=np.int32)
        lens = np.zeros(len(keys), dtype=np.int32)
        for a in range(0, len(keys), chunk):
            texts = [table[int(k)] for k in keys[a:a + chunk]]
            enc = tokenizer(texts, padding=False, truncation="longest_first", max_length=max_len)["input_ids"]
            for j, row in enumerate(enc):
                ids[a + j, :len(row)] = row
                lens[a + j] = len(row)
        return cls(keys, ids, lens)

    def save(self, stem: str, meta: Optional[dict] = None):
        """Each file is written under a temporary name and renamed into place (a reader never sees a half-written array); the
        ``.meta.json`` (max_len, tokenizer, row count) goes last and is what :meth:`load` looks for."""
        for suffix, arr in ((".keys.npy", self.keys), (".ids.npy", self.ids), (".lens.npy", self.lens)):
            tmp = f"{stem}{suffix}.tmp{os.getpid()}.npy"
            np.save(tmp, arr)
            os.replace(tmp, stem + suffix)
        tmp = f"{stem}.meta.json.tmp{os.getpid()}"
        with open(tmp, "w") as fh:
            json.dump({"rows": int(len(self.keys)), "max_len": int(self.ids.shape[1]), **(meta or {})}, fh)
        os.replace(tmp, stem + ".meta.json")

    @staticmethod
    def exists(stem: str) -> bool:
        return os.path.exists(stem + ".meta.json")

    @classmethod
    def load(cls, stem: str, expect: Optional[dict] = None) -> "TokenCache":
        """``expect``: metadata the cache must have been built with (max_len, tokenizer name); a mismatch raises instead of
        silently serving tokens of another tokenizer / truncation."""
        with open(stem + ".meta.json") as fh:
            meta = json.load(fh)
        for k, v in (expect or {}).items():
            if meta.get(k) != v:
                raise ValueError(f"token cache {stem}: built with {k}={meta.get(k)!r}, this run wants {v!r}; delete it or use another --token_cache_dir")
        c = cls(np.load(stem + ".keys.npy"), np.load(stem + ".ids.npy", mmap_mode="r"), np.load(stem + ".lens.npy"))
        if len(c.keys) != meta["rows"] or c.ids.shape != (meta["rows"], meta["max_len"]):
            raise ValueError(f"token cache {stem}: arrays do not match their metadata (truncated write?)")
        return c

    def batch(self, keys, pad_id: int = 0) -> Dict[str, torch.Tensor]:
        """``{'input_ids', 'attention_mask'}`` int64 ``[len(keys), longest]``: what ``tokenizer(texts, padding=True, ...)`` returns."""
        rows = np.fromiter((self._row[int(k)] for k in keys), dtype=np.int64, count=len(keys))
        lens = self.lens[rows]
        width = int(lens.max()) if len(rows) else 0
        ids = np.asarray(self.ids[rows][:, :width], dtype=np.int64)
        mask = (np.arange(width)[None, :] < lens[:, None]).astype(np.int64)
        if pad_id != 0:
            ids = np.where(mask == 1, ids, pad_id)
        return {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}


def attach_lengths(batch):
    """Token counts of the passages, taken from the attention mask while it is on the HOST ("lengths": what lets the encoder pack the
    batch without a device -> host sync; CLDRD_PACK=0 turns packing off).  Only right-padded masks (as HF tokenizers pad) get them.
    numpy on the zero-copy view, not torch: one thread, ~50 us for [256, 128] (torch's intra-op pool on a 256-CPU host: 18 ms for the same sum).
    The collate functions call this in the loader's worker processes; batch_to_device calls it for batches that come from elsewhere."""
    nw = batch.get("nway_passages")
    if not (hasattr(nw, "items") and "lengths" not in nw and isinstance(nw.get("attention_mask"), torch.Tensor)
            and not nw["attention_mask"].is_cuda):
        return batch
    m = nw["attention_mask"].numpy()
    lens = np.count_nonzero(m, axis=-1)
    if not np.array_equal(m != 0, np.arange(m.shape[-1]) < lens[..., None]):
        return batch
    nw = dict(nw.items())
    nw["lengths"] = torch.from_numpy(lens.reshape(-1).astype(np.int64))
    batch = dict(batch.items())
    batch["nway_passages"] = nw
    return batch


class NwayDataset(torch.utils.data.Dataset):
    def __init__(self, qid_to_query, pid_to_passage, train_examples, tokenizer, max_query_len, max_passage_len, label_mode="3",
                 query_cache: Optional[TokenCache] = None, passage_cache: Optional[TokenCache] = None):
        super().__init__()
        self.qid_to_query = qid_to_query
        self.pid_to_passage = pid_to_passage
        self.train_examples = train_examples
        self.tokenizer = tokenizer
        self.max_query_len = max_query_len
        self.max_passage_len = max_passage_len
        self.label_mode = label_mode
        self.query_cache, self.passage_cache = query_cache, passage_cache
        assert self.label_mode in LABEL_MODES
        self._labels = labels_for_mode(label_mode)

    def __len__(self):
        return len(self.train_examples)

    def __getitem__(self, idx):
        ex = self.train_examples[idx]
        rel, neg = ex["relT_pids"], ex["neg_pids"]
        assert len(rel) == _REL[self.label_mode] and len(neg) == len(_NEG[self.label_mode]), \
            f"label mode {self.label_mode} needs {_REL[self.label_mode]} relT and {len(_NEG[self.label_mode])} neg pids"
        item = {"qid": ex["qid"], "relT_pids": rel, "neg_pids": neg, "labels": list(self._labels)}
        if self.query_cache is None:
            item["query"] = self.qid_to_query[ex["qid"]]
            item["relT_passages"] = [self.pid_to_passage[p] for p in rel]
            item["neg_passages"] = [self.pid_to_passage[p] for p in neg]
        return item

    def collate_fn(self, batch):
        qids = np.array([b["qid"] for b in batch], dtype=np.int64)
        rel = np.array([b["relT_pids"] for b in batch], dtype=np.int64)
        neg = np.array([b["neg_pids"] for b in batch], dtype=np.int64).reshape(len(batch), -1)
        bz, nway = len(batch), rel.shape[1] + neg.shape[1]
        nway_pids = np.concatenate((rel, neg), axis=-1)
        if self.query_cache is not None:
            pad = getattr(self.tokenizer, "pad_token_id", 0) or 0
            queries = self.query_cache.batch(qids, pad)
            passages = self.passage_cache.batch(nway_pids.reshape(-1), pad)
        else:
            texts = []
            for b in batch:
                texts += b["relT_passages"] + b["neg_passages"]
            queries = self.tokenizer([b["query"] for b in batch], padding=True, truncation="longest_first", return_tensors="pt",
                                     max_length=self.max_query_len)
            passages = self.tokenizer(texts, padding=True, truncation="longest_first", return_tensors="pt",
                                      max_length=self.max_passage_len)
        passages = {k: v.view(bz, nway, -1) for k, v in passages.items()}
        return attach_lengths({"qid": qids, "relT_pids": rel, "neg_pids": neg, "nway_pids": nway_pids, "query": queries,
                               "nway_passages": passages, "labels": torch.FloatTensor([b["labels"] for b in batch])})

    # ---- constructors (reference :120-470) --------------------------------------------------------------------
    @classmethod
    def _make(cls, queries_path, passages_path, examples, tokenizer, max_query_len, max_passage_len, label_mode, **kw):
        return cls(_read_queries(queries_path), _read_passages(passages_path), examples, tokenizer, max_query_len, max_passage_len,
                   label_mode=label_mode, **kw)

    @classmethod
    def create_from_file(cls, queries_path, passages_path, training_path, tokenizer, max_query_len, max_passage_len, label_mode="3"):
        """one JSON document holding the list of examples (reference :120-146)"""
        with open(training_path, "r") as fh:
            examples = json.load(fh)
        return cls._make(queries_path, passages_path, examples, tokenizer, max_query_len, max_passage_len, label_mode)

    @classmethod
    def dist_create_from_file(cls, queries_path, passages_path, training_path, tokenizer, max_query_len, max_passage_len, rank, nranks):
        """JSON lines, line i on rank i % nranks, default label mode (reference :148-178)"""
        assert nranks > 1 and rank in range(nranks)
        examples = _read_json_lines(training_path, rank, nranks, lambda ex: ex)
        return cls._make(queries_path, passages_path, examples, tokenizer, max_query_len, max_passage_len, "3")

    @classmethod
    def create_from_json_line_file(cls, queries_path, passages_path, training_path, tokenizer, max_query_len, max_passage_len, label_mode):
        """``{"qid", "rel_pid", "neg_pids"}`` per line -> a one-element relT list (reference :180-211)"""
        def convert(ex):
            assert "relT_pids" not in ex and "rel_pid" in ex
            ex["relT_pids"] = [ex.pop("rel_pid")]
            return ex
        examples = _read_json_lines(training_path, -1, None, convert)
        return cls._make(queries_path, passages_path, examples, tokenizer, max_query_len, max_passage_len, label_mode)

    @classmethod
    def create_from_relT_most_semi_hard_file(cls, queries_path, passages_path, training_path, tokenizer, max_query_len, max_passage_len,
                                             label_mode, rank=-1, nranks=None, _modes=None):
        """``{"qid", "relT_pids", "most_hard_pids", "semi_hard_pids"}`` per line; negatives = most hard then semi hard
        (reference :213-258, and the four constructors below with their label-mode guards)"""
        _check_rank(rank, nranks)
        if _modes is not None:
            assert label_mode in _modes
        examples = _read_json_lines(training_path, rank, nranks, _hard_negatives)
        return cls._make(queries_path, passages_path, examples, tokenizer, max_query_len, max_passage_len, label_mode)

    @classmethod
    def create_from_10relT_20neg_file(cls, queries_path, passages_path, training_path, tokenizer, max_query_len, max_passage_len, label_mode,
                                               rank=-1, nranks=None):
        return cls.create_from_relT_most_semi_hard_file(queries_path, passages_path, training_path, tokenizer, max_query_len,
                                                        max_passage_len, label_mode, rank=rank, nranks=nranks, _modes=("3", "9"))

    @classmethod
    def create_from_20relT_10neg_file(cls, queries_path, passages_path, training_path, tokenizer, max_query_len, max_passage_len, label_mode,
                                               rank=-1, nranks=None):
        return cls.create_from_relT_most_semi_hard_file(queries_path, passages_path, training_path, tokenizer, max_query_len,
                                                        max_passage_len, label_mode, rank=rank, nranks=nranks, _modes=("5", "10"))

    @classmethod
    def create_from_30relT_file(cls, queries_path, passages_path, training_path, tokenizer, max_query_len, max_passage_len, label_mode,
                                         rank=-1, nranks=None):
        return cls.create_from_relT_most_semi_hard_file(queries_path, passages_path, training_path, tokenizer, max_query_len,
                                                        max_passage_len, label_mode, rank=rank, nranks=nranks, _modes=("6",))

    @classmethod
    def create_from_5relT_25neg_file(cls, queries_path, passages_path, training_path, tokenizer, max_query_len, max_passage_len, label_mode,
                                              rank=-1, nranks=None):
        return cls.create_from_relT_most_semi_hard_file(queries_path, passages_path, training_path, tokenizer, max_query_len,
                                                        max_passage_len, label_mode, rank=rank, nranks=nranks, _modes=("7", "8"))

    # ---- pre-tokenised cache ------------------------------------------------------------------------------------
    def with_token_cache(self, cache_dir: Optional[str] = None, build: bool = True) -> "NwayDataset":
        """Tokenise every query / passage once (or load ``cache_dir/{queries,passages}.*`` if present) and serve batches from
        the cache from now on.  ``build=False`` (ranks > 0 of a distributed run, after rank 0 has built and a barrier): load only."""
        tok_name = str(getattr(self.tokenizer, "name_or_path", type(self.tokenizer).__name__))

        def get(stem, table, max_len):
            meta = {"max_len": int(max_len), "tokenizer": tok_name}
            if cache_dir and TokenCache.exists(os.path.join(cache_dir, stem)):
                return TokenCache.load(os.path.join(cache_dir, stem), expect=meta)
            if not build:
                raise FileNotFoundError(f"token cache {os.path.join(str(cache_dir), stem)} is missing (rank 0 builds it)")
            c = TokenCache.build(table, self.tokenizer, max_len)
            if cache_dir:
                os.makedirs(cache_dir, exist_ok=True)
                c.save(os.path.join(cache_dir, stem), meta)
            return c
        self.query_cache = get("queries", self.qid_to_query, self.max_query_len)
        self.passage_cache = get("passages", self.pid_to_passage, self.max_passage_len)
        return self
