"""cl-drd_amd/dataset/nway_dataset.py against batches produced by the REFERENCE ``dataset/nway_dataset.py`` on the committed
fixture (tests/golden/make_dataset_golden.py; the reference is not needed to run this test).  Integer arrays and labels must
be identical, for every label mode, for the rank-sharded constructors, and through the pre-tokenised cache."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from toy_tokenizer import make_tokenizer  # noqa: E402

from cldrd_amd.dataset.nway_dataset import LABEL_MODES, NwayDataset, TokenCache, labels_for_mode  # noqa: E402

FIX = os.path.join(HERE, "golden", "nway_dataset_fixture")
G = np.load(os.path.join(HERE, "golden", "nway_dataset.npz"))
Q, C = os.path.join(FIX, "queries.tsv"), os.path.join(FIX, "collection.tsv")
FILES = {"2": "10relT_20neg", "3": "10relT_20neg", "4": "10relT_20neg", "9": "10relT_20neg", "5": "20relT_10neg",
         "10": "20relT_10neg", "6": "30relT", "7": "5relT_25neg", "8": "5relT_25neg"}
CTOR = {"10relT_20neg": "create_from_10relT_20neg_file", "20relT_10neg": "create_from_20relT_10neg_file",
        "30relT": "create_from_30relT_file", "5relT_25neg": "create_from_5relT_25neg_file"}


def check(tag, ds, idxs=None):
    idxs = G[f"{tag}.idxs"].tolist() if idxs is None else idxs
    batch = ds.collate_fn([ds[i] for i in idxs])
    for k in ("qid", "relT_pids", "neg_pids", "nway_pids"):
        assert np.array_equal(np.asarray(batch[k]).reshape(G[f"{tag}.{k}"].shape), G[f"{tag}.{k}"]), (tag, k)
    assert np.array_equal(batch["labels"].numpy(), G[f"{tag}.labels"]), tag
    for side in ("query", "nway_passages"):
        for k in ("input_ids", "attention_mask"):
            got = batch[side][k].numpy()
            assert got.dtype == np.int64 and np.array_equal(got, G[f"{tag}.{side}.{k}"]), (tag, side, k)


def build(mode, **kw):
    stem = FILES[mode]
    name = CTOR[stem] if mode not in ("2", "4") else "create_from_relT_most_semi_hard_file"
    return getattr(NwayDataset, name)(Q, C, os.path.join(FIX, f"train_{stem}.jsonl"), make_tokenizer(), max_query_len=6,
                                      max_passage_len=16, label_mode=mode, **kw)


@pytest.mark.parametrize("mode", sorted(FILES, key=int))
def test_batches_match_the_reference(mode):
    check(f"mode{mode}", build(mode))


def test_mode1_json_line_file():
    ds = NwayDataset.create_from_json_line_file(Q, C, os.path.join(FIX, "train_1rel_5neg.jsonl"), make_tokenizer(), max_query_len=6,
                                                max_passage_len=16, label_mode="1")
    check("mode1", ds)


@pytest.mark.parametrize("mode", ["8", "9"])
def test_rank_sharding_is_line_index_modulo_nranks(mode):
    sizes = []
    for rank in range(3):
        ds = build(mode, rank=rank, nranks=3)
        sizes.append(len(ds))
        check(f"mode{mode}.rank{rank}", ds)
    assert sum(sizes) == 12


@pytest.mark.parametrize("mode", ["3", "8", "6"])
def test_token_cache_gives_identical_batches(mode, tmp_path):
    ds = build(mode).with_token_cache(str(tmp_path))
    check(f"mode{mode}", ds)
    ds2 = build(mode).with_token_cache(str(tmp_path))          # second time: loaded from the .npy files, memory-mapped
    assert isinstance(ds2.passage_cache, TokenCache) and not ds2.passage_cache.ids.flags.writeable
    check(f"mode{mode}", ds2)


def test_label_table():
    assert set(LABEL_MODES) == {str(i) for i in range(1, 11)}
    assert labels_for_mode("9") == list(1.0 / np.arange(1, 11)) + [-0.25] * 10 + [-0.5] * 10
    assert labels_for_mode("4")[:3] == [1.0, 0.9, 0.9] and len(labels_for_mode("6")) == 30
    with pytest.raises(ValueError):
        labels_for_mode("11")
    with pytest.raises(AssertionError):        # wrong example layout for the mode (reference: the per-mode asserts)
        ds = build("3")
        ds.label_mode = "8"
        ds[0]
