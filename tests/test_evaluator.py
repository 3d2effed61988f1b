"""cl-drd_amd/evaluation/retrieval_evaluator.py against the metrics the REFERENCE evaluator produced on the committed fixture
(tests/golden/make_evaluator_golden.py).  Keys and their order identical; values to 1e-12; per-query tables and CSV identical."""
import os

import numpy as np
import pytest

from cldrd_amd.evaluation import RankingEvaluator

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(HERE, "golden", "evaluator_fixture")
G = np.load(os.path.join(HERE, "golden", "evaluator.npz"))


@pytest.mark.parametrize("kind", ["dev", "trec"])
@pytest.mark.parametrize("run", ["run2.tsv", "run3.tsv", "run4.tsv"])
def test_metrics_match_the_reference(kind, run, tmp_path):
    qrels = "qrels.dev.tsv" if kind == "dev" else "qrels.trec.txt"
    ev = RankingEvaluator(os.path.join(FIX, qrels), is_trec=(kind == "trec"))
    out_csv = str(tmp_path / "pq.csv")
    d, (rr, rec, nd) = ev.compute_metrics(os.path.join(FIX, run), return_per_query=True, per_query_metrics_path=out_csv)
    tag = f"{kind}.{run}"
    assert list(d.keys()) == G[tag + ".keys"].tolist()
    assert np.allclose(np.array([float(v) for v in d.values()]), G[tag + ".values"], rtol=1e-12, atol=1e-15)
    assert np.allclose(rr, G[tag + ".rr"], atol=1e-15) and np.allclose(rec, G[tag + ".rec"], atol=1e-15)
    assert np.allclose(nd, G[tag + ".ndcg"], rtol=1e-12, atol=1e-15)
    assert open(out_csv).read() == open(os.path.join(FIX, f"per_query.{kind}.{run}.csv")).read()
    # plain call returns the same dict
    d2 = ev.compute_metrics(os.path.join(FIX, run))
    assert list(d2) == list(d) and all(float(d2[k]) == float(d[k]) for k in d)


def test_custom_cutoffs():
    ev = RankingEvaluator(os.path.join(FIX, "qrels.dev.tsv"), mrr_at_k=[5, 20], ndcg_at_k=[3, 7, 50], recall_at_k=[10, 100, 500],
                          map_at_k=100)
    d = ev.compute_metrics(os.path.join(FIX, "run3.tsv"))
    assert list(d.keys()) == G["custom.keys"].tolist()
    assert np.allclose(np.array([float(v) for v in d.values()]), G["custom.values"], rtol=1e-12, atol=1e-15)


def test_hand_computed_case(tmp_path):
    q = tmp_path / "qrels.tsv"
    q.write_text("1\t0\t10\t1\n1\t0\t11\t1\n2\t0\t20\t1\n3\t0\t30\t0\n")
    r = tmp_path / "run.tsv"
    r.write_text("1\t99\t1\n1\t10\t2\n1\t98\t3\n1\t11\t4\n2\t97\t1\n2\t96\t2\n4\t10\t1\n")
    d = RankingEvaluator(str(q), mrr_at_k=[10], ndcg_at_k=[10], recall_at_k=[1], map_at_k=10).compute_metrics(str(r))
    # query 1: first relevant at rank 2; query 2: no hit; query 4: no qrels (not counted); query 3 has only a zero grade
    assert d["QueriesRanked"] == 2 and d["MRR@10"] == pytest.approx(0.25) and d["QueriesWithRelevant@10"] == 1
    assert d["Recall@1"] == 0.0 and d["MAP@10"] == pytest.approx((1 / 2 + 2 / 4) / 2 / 2)
    ndcg1 = (1 / np.log2(3) + 1 / np.log2(5)) / (1 + 1 / np.log2(3))
    assert d["nDCG@10"] == pytest.approx(ndcg1 / 2)
    with pytest.raises(ValueError):
        bad = tmp_path / "bad.tsv"
        bad.write_text("1\t2\t3\t4\t5\n")
        RankingEvaluator(str(q)).compute_metrics(str(bad))
