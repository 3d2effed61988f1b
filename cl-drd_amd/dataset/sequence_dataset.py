"""Encode-side input contract of the index path (reference ``dataset/sequence_dataset.py:31-55``): batches
``{"seq": {"input_ids", "attention_mask"} int64 [b, L], "id": list[int]}``.

``SequenceDataset`` reads the reference's ``id\\ttext`` TSV and tokenises in ``collate_fn`` with any HuggingFace-style
tokenizer (needs a local vocab: there is no network here).  ``SyntheticSequenceDataset`` produces MSMARCO-shaped token ids
from the portable generator, which is what the benchmarks and tests use."""
from __future__ import annotations

import json
import os
import time

import numpy as np
import torch

from .. import synthetic as syn


class SequenceDataset(torch.utils.data.Dataset):
    def __init__(self, ids, seqs, tokenizer, max_length, is_query=False):
        self.ids, self.seqs, self.tokenizer, self.max_length, self.is_query = ids, seqs, tokenizer, int(max_length), is_query

    @classmethod
    def create_from_seqs_file(cls, path, tokenizer, max_length, is_query=False):
        id_to_seq = {}              # the reference's dict (:31-42): a repeated id keeps its first position and its last text
        with open(path) as fh:
            for line in fh:
                a = line.strip().split("\t")
                if len(a) < 2:
                    continue
                id_to_seq[int(a[0])] = a[1]
        return cls(list(id_to_seq.keys()), list(id_to_seq.values()), tokenizer, max_length, is_query)

    def __len__(self):
        return len(self.ids)

    def __getitem__(self, i):
        return self.ids[i], self.seqs[i]

    def collate_fn(self, batch):
        ids, seqs = zip(*batch)
        enc = self.tokenizer(list(seqs), padding=True, truncation="longest_first", max_length=self.max_length, return_tensors="pt")
        return {"seq": {"input_ids": enc["input_ids"], "attention_mask": enc["attention_mask"]}, "id": list(ids)}


class SequenceTokenCache:
    """The ``id\\ttext`` collection tokenised ONCE (for the index path): ``ids [n, max_length]`` uint16 (int32 when
    the vocabulary needs it; zero padded), ``lens [n]`` int32, ``keys [n]`` int64, three ``.npy`` files that are memory-mapped on load.

    The reference tokenises every batch of 512 passages inside ``collate_fn`` (dataset/sequence_dataset.py:44-55) with four DataLoader
    workers, and every rank first reads the whole 8.8 M-line TSV into a dict (:31-42): a few thousand passages per second per worker
    against an encoder that takes ~75 000 per second.  Built by one streaming pass over the TSV (rows in file order, exactly the rows
    ``create_from_seqs_file`` keeps, a repeated id keeping its first position and last text as the reference's dict does); afterwards a
    rank maps the files and touches only the pages of its own row range."""

    def __init__(self, keys, ids, lens, meta):
        self.keys, self.ids, self.lens, self.meta = keys, ids, lens, meta

    def __len__(self):
        return int(self.keys.shape[0])

    @staticmethod
    def stem_for(cache_dir: str, path: str, max_length: int) -> str:
        return os.path.join(cache_dir, f"{os.path.basename(path)}.L{int(max_length)}.seqcache")

    @staticmethod
    def source_meta(path: str, tokenizer, max_length: int) -> dict:
        st = os.stat(path)
        return {"source": os.path.abspath(path), "source_bytes": int(st.st_size), "source_mtime_ns": int(st.st_mtime_ns),
                "max_length": int(max_length), "tokenizer": str(getattr(tokenizer, "name_or_path", "") or type(tokenizer).__name__),
                "vocab_size": int(len(tokenizer))}

    @classmethod
    def build(cls, path: str, tokenizer, max_length: int, stem: str, chunk: int = 8192) -> "SequenceTokenCache":
        meta = cls.source_meta(path, tokenizer, max_length)
        # pass 1: the rows create_from_seqs_file keeps, in the order its dict would hold them
        order, last = {}, []
        with open(path) as fh:
            for ln, line in enumerate(fh):
                a = line.strip().split("\t")
                if len(a) != 2:
                    if not line.strip():
                        continue
                    raise ValueError(f"{path}:{ln + 1}: expected 'id<TAB>text' (the reference unpacks exactly two fields)")
                k = int(a[0])
                if k in order:
                    last[order[k]] = ln
                else:
                    order[k] = len(last)
                    last.append(ln)
        n = len(last)
        keep = {ln: row for row, ln in enumerate(last)}
        dtype = np.uint16 if meta["vocab_size"] <= 65536 else np.int32
        os.makedirs(os.path.dirname(stem) or ".", exist_ok=True)
        tmp = f".tmp{os.getpid()}"
        ids = np.lib.format.open_memmap(stem + ".ids.npy" + tmp, mode="w+", dtype=dtype, shape=(n, int(max_length)))
        lens = np.zeros(n, dtype=np.int32)
        keys = np.zeros(n, dtype=np.int64)

        def flush(rows, texts):
            enc = tokenizer(texts, padding=False, truncation="longest_first", max_length=int(max_length))["input_ids"]
            for r, t in zip(rows, enc):
                ids[r, :len(t)] = t
                lens[r] = len(t)

        rows, texts = [], []
        with open(path) as fh:      # pass 2: tokenise chunk by chunk (never the whole collection in memory)
            for ln, line in enumerate(fh):
                row = keep.get(ln)
                if row is None:
                    continue
                a = line.strip().split("\t")
                keys[row] = int(a[0])
                rows.append(row)
                texts.append(a[1])
                if len(rows) >= chunk:
                    flush(rows, texts)
                    rows, texts = [], []
        if rows:
            flush(rows, texts)
        ids.flush()
        del ids
        os.replace(stem + ".ids.npy" + tmp, stem + ".ids.npy")
        for suffix, arr in ((".lens.npy", lens), (".keys.npy", keys)):
            np.save(stem + suffix + tmp + ".npy", arr)
            os.replace(stem + suffix + tmp + ".npy", stem + suffix)
        meta["rows"] = n
        with open(stem + ".meta.json" + tmp, "w") as fh:
            json.dump(meta, fh)
        os.replace(stem + ".meta.json" + tmp, stem + ".meta.json")        # last: what load() looks for
        return cls.load(stem, meta)

    @classmethod
    def load(cls, stem: str, expect: dict | None = None) -> "SequenceTokenCache":
        with open(stem + ".meta.json") as fh:
            meta = json.load(fh)
        for k, v in (expect or {}).items():
            if k != "rows" and meta.get(k) != v:
                raise ValueError(f"sequence token cache {stem}: built with {k}={meta.get(k)!r}, this run has {v!r}: delete it or use another --token_cache_dir")
        ids = np.load(stem + ".ids.npy", mmap_mode="r")
        lens, keys = np.load(stem + ".lens.npy", mmap_mode="r"), np.load(stem + ".keys.npy", mmap_mode="r")
        if ids.shape != (meta["rows"], meta["max_length"]) or lens.shape[0] != meta["rows"] or keys.shape[0] != meta["rows"]:
            raise ValueError(f"sequence token cache {stem}: arrays do not match their metadata (truncated write?)")
        return cls(keys, ids, lens, meta)

    @classmethod
    def open_or_build(cls, cache_dir: str, path: str, tokenizer, max_length: int, rank: int = 0, world: int = 1, wait_s: float = 7200.0):
        """Rank 0 builds a missing / stale cache (atomic renames); the other ranks of an index_text run - independent processes, no
        process group on that path - wait for its metadata file to appear."""
        stem = cls.stem_for(cache_dir, path, max_length)
        want = cls.source_meta(path, tokenizer, max_length)
        stale_err = None
        if os.path.exists(stem + ".meta.json"):
            try:
                return cls.load(stem, want)
            except ValueError as exc:
                stale_err = exc            # stale (another collection / tokenizer / length): rank 0 rebuilds it below, the others wait for that
        marker = stem + ".building"
        if rank == 0:
            # a marker file FIRST: it tells the waiting ranks that rank 0 has decided to (re)build - from then on a stale metadata file is
            # rank 0's business and no reason to give up.  Then the metadata file goes: waiters key on it, and while a rebuild replaces
            # ids / lens / keys one by one nobody may load a mix of new arrays and old metadata
            with open(marker, "w") as fh:
                fh.write(str(os.getpid()))
            try:
                os.remove(stem + ".meta.json")
            except FileNotFoundError:
                pass
            try:
                return cls.build(path, tokenizer, max_length, stem)
            finally:
                try:
                    os.remove(marker)
                except FileNotFoundError:
                    pass
        # Other ranks wait for rank 0 - but not blindly: a STALE metadata file with no build marker next to it for `stale_grace_s` in a row
        # means rank 0 does not see it as stale (another tokenizer / collection there), has not reached this point yet (model load, query
        # tokenisation: hence ten minutes, not two), or died: fail with the reason instead of polling for two hours.  The clock only runs
        # while neither the marker nor a fresh metadata file exists; progress of a build is visible as the growing temporary ids file.
        stale_grace_s = min(wait_s, 600.0)
        t0 = time.time()
        stale_since = None
        last_err = stale_err
        while time.time() - t0 < wait_s:
            if os.path.exists(stem + ".meta.json"):
                try:
                    return cls.load(stem, want)
                except ValueError as exc:
                    last_err = exc
                    if os.path.exists(marker):
                        stale_since = None
                    else:
                        stale_since = time.time() if stale_since is None else stale_since
                        if time.time() - stale_since > stale_grace_s:
                            raise ValueError(f"{exc} (rank {rank}: rank 0 has not started to replace this cache within {stale_grace_s:.0f} s - "
                                             f"does it run with the same collection / tokenizer / max_length?)") from exc
            else:
                stale_since = None
            time.sleep(1.0)
        raise TimeoutError(f"sequence token cache {stem}: rank 0 did not finish it within {wait_s:.0f} s"
                           + (f" (last problem: {last_err})" if last_err else ""))


class CachedSequenceDataset(torch.utils.data.Dataset):
    """Whole encode batches of rows [lo, hi) of a :class:`SequenceTokenCache` with the ``collate_fn`` layout of the reference
    (dataset/sequence_dataset.py:44-55): ``input_ids`` / ``attention_mask`` int64 padded to the longest row OF THE BATCH, ``id`` list[int].

    ``bucket_window`` > 0 (round 6; retriever.index_text turns it on): the batches are no longer 512 consecutive rows but LENGTH BUCKETS.  The
    cache knows every row's token count, so inside each window of ``bucket_window`` consecutive rows the rows are sorted by length and cut
    into chunks whose padded size (rows x longest row) is at most ``token_budget`` (and ``max_rows`` rows): a chunk's rows have about the same
    length - its padded layout, which the attention kernels work on, is nearly all real tokens - and its row count sits just under 256 x 256,
    so the Linear layers run whole rounds of 256-row tiles on the 256 CUs.  A passage's embedding does not depend on what it is batched with; every batch
    carries ``"row"`` (positions in [0, hi - lo)) and ``get_embeddings_from_scratch`` puts the rows back in collection order.  Measured on the
    full MS MARCO-shaped collection: DESIGN.md section 6."""

    def __init__(self, cache: SequenceTokenCache, lo: int = 0, hi: int | None = None, batch_size: int = 512, pad_id: int = 0,
                 bucket_window: int = 0, token_budget: int = 65536, max_rows: int = 2048):
        self.cache, self.lo, self.hi = cache, int(lo), int(len(cache) if hi is None else hi)
        self.batch_size, self.pad_id = int(batch_size), int(pad_id)
        self.n_rows = self.hi - self.lo          # get_embeddings_from_scratch allocates its [n, D] result once when a dataset says this
        self.chunks = None
        if bucket_window > 0 and self.n_rows > 0:
            self.chunks = self.length_buckets(np.asarray(cache.lens[self.lo:self.hi]), int(bucket_window), int(token_budget), int(max_rows))

    @staticmethod
    def length_buckets(lens, window: int, token_budget: int, max_rows: int):
        """[int64 arrays of row positions]: per window of consecutive rows, rows in (length, position) order cut greedily so that a chunk's PADDED
        size - rows x its longest row, what the encoder computes on when a batch is too full to be worth packing - stays within the budget"""
        chunks = []
        lens = np.maximum(np.asarray(lens, dtype=np.int64), 1)
        for w0 in range(0, lens.shape[0], window):
            wl = lens[w0:w0 + window]
            order = np.argsort(wl, kind="stable")
            sl = wl[order]
            start = 0
            while start < order.shape[0]:
                cand = sl[start:start + max_rows]
                fits = np.nonzero(np.arange(1, cand.shape[0] + 1) * cand <= token_budget)[0]      # ascending lengths: rows x longest is monotone
                end = start + (int(fits[-1]) + 1 if fits.size else 1)
                chunks.append(np.sort(order[start:end]) + w0)       # ascending positions: sequential reads of the memory map
                start = end
        return chunks

    def __len__(self):
        if self.chunks is not None:
            return len(self.chunks)
        return (self.hi - self.lo + self.batch_size - 1) // self.batch_size

    def _batch(self, rows_abs, lens):
        width = int(lens.max()) if lens.shape[0] else 0
        ids = np.asarray(self.cache.ids[rows_abs, :width] if not isinstance(rows_abs, slice) else self.cache.ids[rows_abs, :width]).astype(np.int64)
        mask = (np.arange(width)[None, :] < lens[:, None]).astype(np.int64)
        if self.pad_id != 0:
            ids = np.where(mask == 1, ids, self.pad_id)
        # "lengths" (host-side token counts): what lets the encoder pack the batch; given here so that nobody has to derive them from the mask
        return {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask), "lengths": lens.astype(np.int64).tolist()}

    def __getitem__(self, b):
        if self.chunks is not None:
            rows = self.chunks[b]
            rows_abs = rows + self.lo
            lens = np.asarray(self.cache.lens[rows_abs])
            return {"seq": self._batch(rows_abs, lens), "id": np.asarray(self.cache.keys[rows_abs]).tolist(), "row": rows.tolist()}
        a = self.lo + b * self.batch_size
        z = min(self.hi, a + self.batch_size)
        lens = np.asarray(self.cache.lens[a:z])
        return {"seq": self._batch(slice(a, z), lens), "id": np.asarray(self.cache.keys[a:z]).tolist()}

    def loader(self, num_workers: int = 2, pin_memory: bool = False):
        """whole batches (already collated); ``pin_memory``: the loader's pinning thread stages them, so the H2D copies of the encode loop are
        asynchronous (retriever.index_text uses it)"""
        kw = dict(prefetch_factor=4, persistent_workers=False) if num_workers > 0 else {}
        return torch.utils.data.DataLoader(self, batch_size=None, shuffle=False, num_workers=num_workers, pin_memory=bool(pin_memory) and torch.cuda.is_available(), **kw)


class SyntheticSequenceDataset(torch.utils.data.Dataset):
    """n MSMARCO-shaped sequences of up to ``max_length`` tokens: rows [first_id, first_id + n) of one endless collection whose row r
    depends on (seed, r) only - shards of it (``first_id`` = the shard's first row) hold exactly the rows a single process would."""

    def __init__(self, n, max_length, seed=99, vocab=syn.VOCAB, ragged=True, first_id=0, batch_size=512):
        self.n, self.max_length, self.seed, self.vocab, self.ragged = int(n), int(max_length), seed, vocab, ragged
        self.first_id, self.batch_size = first_id, batch_size
        self.n_rows = self.n

    def __len__(self):
        return (self.n + self.batch_size - 1) // self.batch_size

    def __getitem__(self, b):
        lo = b * self.batch_size
        rows = min(self.batch_size, self.n - lo)
        batch = syn.seq_rows(self.seed, self.first_id + lo, rows, self.max_length, vocab=self.vocab, ragged=self.ragged)
        if self.ragged:      # pad to the longest sequence of the batch, as the HF tokenizer does (padding=True)
            longest = int(batch["seq"]["attention_mask"].sum(1).max())
            batch["seq"] = {k: v[:, :longest].contiguous() for k, v in batch["seq"].items()}
        return batch

    def loader(self):
        """Iterate whole batches (already collated)."""
        return torch.utils.data.DataLoader(self, batch_size=None, shuffle=False, num_workers=0)
