"""Index-path input contract (reference dataset/sequence_dataset.py:31-55) against batches the REFERENCE's SequenceDataset produced
from the fixture collection with the toy tokenizer (tests/golden/make_dataset_golden.py -> sequence_dataset.npz): the tokenise-per-batch
dataset and the tokenise-once memory-mapped cache must both reproduce them, for the whole collection and for per-rank row ranges."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import GOLDEN  # noqa: E402
from toy_tokenizer import make_tokenizer  # noqa: E402

from cldrd_amd.dataset import CachedSequenceDataset, SequenceDataset, SequenceTokenCache  # noqa: E402
from cldrd_amd.retriever.index_text import collection_loader  # noqa: E402
from cldrd_amd.retriever.retrieval_utils import ShardedFlatIPIndex  # noqa: E402

COLLECTION = os.path.join(GOLDEN, "nway_dataset_fixture", "collection.tsv")


def golden_batches():
    g = np.load(os.path.join(GOLDEN, "sequence_dataset.npz"))
    n = int(g["n"])
    return n, [(g[f"b{b}.input_ids"], g[f"b{b}.attention_mask"], g[f"b{b}.id"].tolist()) for b in range((n + 6) // 7)]


def same(batch, want):
    ids, mask, keys = want
    return (np.array_equal(batch["seq"]["input_ids"].numpy(), ids) and np.array_equal(batch["seq"]["attention_mask"].numpy(), mask)
            and batch["id"] == keys and batch["seq"]["input_ids"].dtype == torch.int64 and isinstance(batch["id"][0], int))


def test_tokenise_per_batch_dataset_matches_the_reference():
    n, gold = golden_batches()
    ds = SequenceDataset.create_from_seqs_file(COLLECTION, make_tokenizer(), 12, is_query=False)
    assert len(ds) == n
    for b, want in enumerate(gold):
        assert same(ds.collate_fn([ds[i] for i in range(7 * b, min(n, 7 * b + 7))]), want), b


def test_token_cache_matches_the_reference_and_reloads(tmp_path):
    n, gold = golden_batches()
    tok = make_tokenizer()
    cache = SequenceTokenCache.open_or_build(str(tmp_path), COLLECTION, tok, 12)
    assert len(cache) == n and cache.ids.dtype == np.uint16 and cache.ids.shape == (n, 12)
    ds = CachedSequenceDataset(cache, batch_size=7)
    assert len(ds) == len(gold)
    for b, want in enumerate(gold):
        assert same(ds[b], want), b
    # second open: loaded (memory-mapped), not rebuilt
    stem = SequenceTokenCache.stem_for(str(tmp_path), COLLECTION, 12)
    t0 = os.stat(stem + ".ids.npy").st_mtime_ns
    again = SequenceTokenCache.open_or_build(str(tmp_path), COLLECTION, tok, 12)
    assert os.stat(stem + ".ids.npy").st_mtime_ns == t0 and isinstance(again.ids, np.memmap)
    # another truncation length is another cache; stale metadata is refused, not served
    with pytest.raises(ValueError):
        SequenceTokenCache.load(stem, SequenceTokenCache.source_meta(COLLECTION, tok, 13))


def test_rank_row_ranges_of_the_cache_cover_the_collection(tmp_path):
    """index_text.py under RANK / WORLD_SIZE: rank r encodes rows shard_bounds(n, world, r); together the ranks see every row once, in
    order, and a rank's batches equal the corresponding slice of the tokenise-per-batch path."""
    tok = make_tokenizer()
    n = len(SequenceTokenCache.open_or_build(str(tmp_path), COLLECTION, tok, 12))
    seen = []
    for r in range(3):
        lo, hi = ShardedFlatIPIndex.shard_bounds(n, 3, r)
        cached = list(collection_loader(COLLECTION, tok, 12, False, str(tmp_path), r, 3))
        plain = list(collection_loader(COLLECTION, tok, 12, False, "", r, 3))
        assert len(cached) == len(plain) == (1 if hi > lo else 0)
        for a, b in zip(cached, plain):
            assert torch.equal(a["seq"]["input_ids"], b["seq"]["input_ids"]) and torch.equal(a["seq"]["attention_mask"], b["seq"]["attention_mask"])
            assert a["id"] == b["id"]
            seen += a["id"]
    assert seen == list(range(n))


def test_repeated_ids_follow_the_reference_dict(tmp_path):
    p = tmp_path / "dup.tsv"
    p.write_text("5\talpha beta\n7\tgamma\n5\tdelta epsilon zeta\n\n9\tpi\n")
    tok = make_tokenizer()
    ds = SequenceDataset.create_from_seqs_file(str(p), tok, 8)
    cache = SequenceTokenCache.build(str(p), tok, 8, str(tmp_path / "c"))
    assert ds.ids == [5, 7, 9] and list(cache.keys) == [5, 7, 9]
    b0, b1 = ds.collate_fn([ds[i] for i in range(3)]), CachedSequenceDataset(cache, batch_size=3)[0]
    assert torch.equal(b0["seq"]["input_ids"], b1["seq"]["input_ids"]) and b1["id"] == [5, 7, 9]
    assert int(b1["seq"]["attention_mask"][0].sum()) == 5          # id 5 carries its LAST text: [CLS] delta epsilon zeta [SEP]


def test_waiting_rank_fails_fast_on_a_stale_cache_that_rank0_never_replaces(tmp_path, monkeypatch):
    """a rank != 0 that finds a cache built with other metadata waits for rank 0's rebuild - but only for a grace period when
    the stale metadata file never goes away (rank 0 not started / sees another tokenizer): ValueError with the reason, not a 2-hour poll.
    And rank 0 removes the metadata file FIRST when it rebuilds, so that no waiter can load new arrays with old metadata."""
    import json
    import time as _time
    tok = make_tokenizer()
    cache = SequenceTokenCache.open_or_build(str(tmp_path), COLLECTION, tok, 12)
    stem = SequenceTokenCache.stem_for(str(tmp_path), COLLECTION, 12)
    del cache
    meta = json.load(open(stem + ".meta.json"))
    stale = dict(meta)
    key = next(k for k in meta if k not in ("rows", "max_length"))
    stale[key] = "something else"
    json.dump(stale, open(stem + ".meta.json", "w"))
    clock = {"t": 1000.0}
    monkeypatch.setattr(_time, "time", lambda: clock["t"])
    monkeypatch.setattr(_time, "sleep", lambda s: clock.__setitem__("t", clock["t"] + 30.0))
    with pytest.raises(ValueError, match="rank 0 has not started to replace"):
        SequenceTokenCache.open_or_build(str(tmp_path), COLLECTION, tok, 12, rank=1, world=2, wait_s=7200.0)
    assert clock["t"] - 1000.0 <= 700.0           # gave up after the grace period (10 min without a build marker), not after wait_s
    # once rank 0 has dropped its ".building" marker the stale metadata is its business - the waiter keeps waiting however late
    # rank 0 arrived, and only the overall wait_s ends the poll
    open(stem + ".building", "w").write("1")
    clock["t"] = 1000.0
    with pytest.raises(TimeoutError):
        SequenceTokenCache.open_or_build(str(tmp_path), COLLECTION, tok, 12, rank=1, world=2, wait_s=3000.0)
    assert clock["t"] - 1000.0 >= 3000.0
    os.remove(stem + ".building")
    monkeypatch.undo()
    # rank 0: the stale metadata disappears before anything else is replaced
    seen = {}
    real_build = SequenceTokenCache.build.__func__

    def spy(cls, *a, **k):
        seen["meta_present_at_build"] = os.path.exists(stem + ".meta.json")
        seen["marker_at_build"] = os.path.exists(stem + ".building")
        return real_build(cls, *a, **k)
    monkeypatch.setattr(SequenceTokenCache, "build", classmethod(spy))
    again = SequenceTokenCache.open_or_build(str(tmp_path), COLLECTION, tok, 12, rank=0, world=2)
    assert seen["meta_present_at_build"] is False and len(again) == meta["rows"]
    assert seen["marker_at_build"] is True and not os.path.exists(stem + ".building")          # the marker lives exactly as long as the build


def test_length_buckets_cover_every_row_once_within_the_token_budget(tmp_path):
    """CachedSequenceDataset(bucket_window=...): inside each window the rows are cut into chunks by length - every row of [lo, hi) exactly once,
    chunk padded sizes (rows x longest row) within the budget, rows of a chunk within a narrow length band, chunks confined to their window, positions ascending;
    each batch carries the collate layout, its token counts and its row positions."""
    rng = np.random.default_rng(5)
    n, L = 5000, 64
    lens = np.clip(np.round(rng.lognormal(3.0, 0.5, n)), 1, L).astype(np.int32)
    ids = np.zeros((n, L), dtype=np.uint16)
    for r in range(n):
        ids[r, :lens[r]] = rng.integers(5, 500, lens[r])
    keys = (np.arange(n, dtype=np.int64) * 7 + 3)
    cache = SequenceTokenCache(keys, ids, lens, {"rows": n, "max_length": L})
    lo, hi = 100, 4700
    ds = CachedSequenceDataset(cache, lo, hi, batch_size=512, bucket_window=1024, token_budget=4096, max_rows=300)
    assert ds.n_rows == hi - lo and len(ds) == len(ds.chunks) > (hi - lo) // 512
    seen = np.zeros(hi - lo, dtype=np.int64)
    for i, rows in enumerate(ds.chunks):
        seen[rows] += 1
        cl = lens[lo:hi][rows]
        assert (len(rows) * cl.max() <= 4096 or len(rows) == 1) and len(rows) <= 300 and np.all(np.diff(rows) > 0)
        assert rows.min() // 1024 == rows.max() // 1024                      # one window
        b = ds[i]
        assert b["row"] == rows.tolist() and b["id"] == keys[rows + lo].tolist() and b["seq"]["lengths"] == cl.tolist()
        w = int(cl.max())
        assert tuple(b["seq"]["input_ids"].shape) == (len(rows), w)
        assert torch.equal(b["seq"]["input_ids"], torch.from_numpy(ids[rows + lo, :w].astype(np.int64)))
        assert torch.equal(b["seq"]["attention_mask"].sum(1), torch.from_numpy(cl.astype(np.int64)))
    assert np.all(seen == 1)
    # the buckets are tight: the padded layout of a chunk is mostly real tokens (the plain 512-row batches of this length mix: ~45 %)
    fill = np.mean([lens[lo:hi][r].sum() / (len(r) * lens[lo:hi][r].max()) for r in ds.chunks if len(r) > 8])
    plain = np.mean([lens[a:a + 512].sum() / (min(512, hi - a) * lens[a:min(hi, a + 512)].max()) for a in range(lo, hi, 512)])
    assert fill > 0.8 and plain < 0.5, (fill, plain)
    # a row longer than the budget still gets a chunk of its own
    ds2 = CachedSequenceDataset(cache, 0, 50, bucket_window=50, token_budget=8)
    assert sorted(int(r) for c in ds2.chunks for r in c) == list(range(50))
