import os

import numpy as np
import pytest

from oracle import optim_ref as O
from oracle import retrieval_ref as R
import cldrd_amd.synthetic as syn

from conftest import GOLDEN


def test_lr_schedule_matches_transformers():
    g = np.load(os.path.join(GOLDEN, "lr_schedule.npz"))
    for name in ("a", "b", "c"):
        warm, total = int(g[name + "/warmup"]), int(g[name + "/total"])
        for s, f in zip(g[name + "/steps"], g[name + "/factor"]):
            assert O.linear_schedule_factor(int(s), warm, total) == pytest.approx(float(f), abs=1e-12)


def test_adamw_step_matches_torch_for_zero_decay():
    """With weight_decay = 0 the legacy HF rule differs from torch.optim.AdamW only in where eps enters;
    check against a hand-rolled float64 formula on a tiny vector and the decay term separately."""
    p, g = np.array([1.0, -2.0, 0.5]), np.array([0.1, -0.3, 0.2])
    m = v = np.zeros(3)
    p1, m1, v1 = O.adamw_step(p, g, m, v, lr=1e-3, step=1, weight_decay=0.0)
    # step 1: m = .1 g, v = .001 g^2, step_size = lr*sqrt(.001)/.1 -> update = lr * g/(|g| + eps/sqrt(.001)) approx
    expect = p - 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9) * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-8)
    assert np.allclose(p1, expect, rtol=0, atol=1e-15)
    p2, _, _ = O.adamw_step(p, g, m, v, lr=1e-3, step=1, weight_decay=0.01)
    assert np.allclose(p2, p1 * (1 - 1e-3 * 0.01), atol=1e-15)


def test_no_decay_rule():
    assert O.no_decay("query_encoder.embeddings.LayerNorm.weight")
    assert O.no_decay("query_encoder.transformer.layer.0.ffn.lin1.bias")
    assert not O.no_decay("query_encoder.transformer.layer.0.sa_layer_norm.weight")   # decayed
    assert not O.no_decay("query_encoder.transformer.layer.0.ffn.lin1.weight")


def test_clip_coef():
    total, coef = O.clip_coef([np.array([3.0]), np.array([4.0])], 1.0)
    assert total == pytest.approx(5.0) and coef == pytest.approx(1.0 / (5.0 + 1e-6))
    assert O.clip_coef([np.array([0.1])], 1.0)[1] == 1.0


def test_flat_ip_search_matches_bruteforce_and_ties():
    emb = syn.corpus_embeddings(7, 3000, 32)
    emb[100] = emb[50]          # exact duplicate row -> tie broken by lower position
    q = syn.corpus_embeddings(8, 5, 32)
    ids = np.arange(3000, dtype=np.int64) * 7 + 3
    D, I = R.flat_ip_search(emb, ids, q, 10, chunk=512)
    full = (q.astype(np.float64) @ emb.astype(np.float64).T).astype(np.float32)
    for r in range(5):
        order = np.lexsort((np.arange(3000), -full[r].astype(np.float64)))[:10]
        assert np.array_equal(I[r], ids[order])
        assert np.array_equal(D[r], full[r][order])
    assert np.all(np.diff(D, axis=1) <= 0)
    # k > ntotal pads with -1
    D2, I2 = R.flat_ip_search(emb[:4], None, q, 6)
    assert np.all(I2[:, 4:] == -1) and np.all(np.isneginf(D2[:, 4:]))


def test_index_retrieve_batches_and_run_file():
    emb = syn.corpus_embeddings(7, 500, 16)
    q = syn.corpus_embeddings(8, 7, 16)
    idx = R.FlatIPIndex(emb, np.arange(500) + 1000)
    s_all, i_all = R.index_retrieve(idx, q, 5, batch=None)
    s_b, i_b = R.index_retrieve(idx, q, 5, batch=3)
    assert np.array_equal(np.array(i_b), i_all) and np.allclose(np.array(s_b), s_all)
    lines = R.run_file_lines([11, 12], i_b[:2], s_b[:2])
    assert lines[0].split("\t")[:3] == ["11", str(i_b[0][0]), "1"] and len(lines) == 10


def test_merge_shards_equals_global():
    emb = syn.corpus_embeddings(9, 1000, 16)
    q = syn.corpus_embeddings(10, 4, 16)
    D, I = R.flat_ip_search(emb, None, q, 20)
    parts = [R.flat_ip_search(emb[lo:lo + 250], np.arange(lo, lo + 250), q, 20) for lo in range(0, 1000, 250)]
    Dm, Im = R.merge_shard_results([p[0] for p in parts], [p[1] for p in parts], 20)
    assert np.array_equal(Im, I) and np.array_equal(Dm, D)


def test_faiss_binary_layout_round_trip_and_hand_assembled_bytes(tmp_path):
    """IndexIDMap(IndexFlatIP) file layout (restated from faiss' index_write.cpp; faiss itself is absent -> unpinned):
    the writer produces exactly the hand-assembled byte string, the reader inverts it, read_index auto-detects the format."""
    import struct
    from cldrd_amd.retriever import retrieval_utils as RU
    emb = (np.arange(12, dtype=np.float32).reshape(4, 3) - 5.0) / 4.0
    ids = np.array([7, 8, 100, 3], dtype=np.int64)
    idx = RU.FlatIPIndex(3)
    idx.add_with_ids(emb, ids)
    path = str(tmp_path / "small.index")
    RU.write_index(idx, path, faiss_format=True)
    hdr = struct.pack("<i", 3) + struct.pack("<q", 4) + struct.pack("<q", 1 << 20) * 2 + b"\x01" + struct.pack("<i", 0)
    want = b"IxMp" + hdr + b"IxFI" + hdr + struct.pack("<Q", 12) + emb.tobytes() + struct.pack("<Q", 4) + ids.tobytes()
    assert open(path, "rb").read() == want
    back = RU.read_index(path)
    assert back.ntotal == 4 and back.d == 3 and np.array_equal(np.asarray(back.embeddings), emb) and np.array_equal(back.ids, ids)
    # a bare IndexFlatIP file (no id map)
    bare = str(tmp_path / "bare.index")
    open(bare, "wb").write(b"IxFI" + hdr + struct.pack("<Q", 12) + emb.tobytes())
    b2 = RU.read_index(bare)
    assert b2.ntotal == 4 and np.array_equal(np.asarray(b2.embeddings), emb)
    with pytest.raises(ValueError):
        open(bare, "wb").write(b"IxFI" + hdr[:-4] + struct.pack("<i", 1) + struct.pack("<Q", 12) + emb.tobytes())     # L2 metric
        RU.read_index(bare)
