"""Ranking metrics of a run file against qrels with the semantics of reference ``evaluation/retrieval_evaluator.py:14-221``
(``RankingEvaluator``): MRR@{10,1000}, nDCG@{10,100}, Recall@{50,1000}, MAP@1000 — the numbers of the reference's README.

Host-side numpy by design (a CPU metric, out of scope for acceleration; section 8f row 1 asks for the
same semantics so that "MRR@10 within +-0.002" can be checked on real MS MARCO).  Semantics kept, quirks included:

* qrels lines ``qid _ pid grade`` split on tab (dev) or single space (``is_trec``); grades <= 1e-5 are dropped (:19-33);
* the run file is taken as already sorted per query, 2-4 tab-separated columns, first two = qid, pid (:44-63);
* a document is binary-relevant if grade >= 1 (dev) or >= 2 (TREC-DL) — MRR, Recall, MAP use that; nDCG uses every graded
  document with DCG = sum grade / log2(1 + rank) and the ideal DCG over the query's sorted grades (:98-176);
* every mean divides by the number of ranked queries that have qrels ("QueriesRanked"), queries without a relevant hit count as 0;
* the per-query reciprocal-rank table has ``len(recall_at_k)`` rows (:89) — kept, so ``len(mrr_at_k) <= len(recall_at_k)`` is
  required exactly as in the reference.

Pinned against the reference on a synthetic run / qrels fixture: tests/test_evaluator.py.
"""
from __future__ import annotations

import csv
from typing import Dict, List

import numpy as np


class RankingEvaluator:
    def __init__(self, qrel_path: str, mrr_at_k: List[int] = (10, 1000), ndcg_at_k: List[int] = (10, 100),
                 recall_at_k: List[int] = (50, 1000), map_at_k: int = 1000, show_progress_bar: bool = False, is_trec: bool = False):
        self.qid_to_relevant_data: Dict[int, Dict[int, float]] = {}
        sep = " " if is_trec else "\t"
        with open(qrel_path, "r") as fh:
            for line in fh:
                qid, _, pid, grade = line.strip().split(sep)
                if float(grade) <= 0.00001:
                    continue
                self.qid_to_relevant_data.setdefault(int(qid), {})[int(pid)] = float(grade)
        self.mrr_at_k, self.ndcg_at_k, self.recall_at_k = list(mrr_at_k), list(ndcg_at_k), list(recall_at_k)
        self.map_at_k = map_at_k
        self.show_progress_bar = show_progress_bar
        self.is_trec = is_trec

    # -----------------------------------------------------------------------------------------------------------
    @staticmethod
    def read_run(ranking_path: str) -> Dict[int, List[int]]:
        ranking: Dict[int, List[int]] = {}
        with open(ranking_path, "r") as fh:
            for line in fh:
                cols = line.strip().split("\t")
                if not 2 <= len(cols) <= 4:
                    raise ValueError("array length is not legal.")
                ranking.setdefault(int(cols[0]), []).append(int(cols[1]))
        return ranking

    def compute_metrics(self, ranking_path, return_per_query=False, per_query_metrics_path=None):
        ranking = self.read_run(ranking_path)
        point = 2.0 if self.is_trec else 1.0
        if not return_per_query:
            return self._calculate_metrics_plain(ranking, self.qid_to_relevant_data, binarization_point=point)
        local, rr, rec, ndcg, qidx_to_qid, qrels = self._calculate_metrics_plain(ranking, self.qid_to_relevant_data,
                                                                              binarization_point=point, return_per_query=True)
        if per_query_metrics_path is not None:
            self._output_per_query_metrics(qidx_to_qid, qrels, per_query_metrics_path, rr, rec, ndcg)
        return local, (rr, rec, ndcg)

    def _calculate_metrics_plain(self, ranking, qrels, binarization_point=1.0, return_per_query=False):
        nq = len(ranking)
        qidx_to_qid = dict(enumerate(ranking))
        rr = np.zeros((len(self.recall_at_k), nq))          # sized by recall_at_k, as the reference (:89)
        rec = np.zeros((len(self.recall_at_k), nq))
        ndcg = np.zeros((len(self.ndcg_at_k), nq))
        ap = np.zeros(nq)
        evaluated = 0
        for qi, (qid, docs) in enumerate(ranking.items()):
            if qid not in qrels:
                continue
            evaluated += 1
            rel_ids = np.array(list(qrels[qid].keys()))
            grades = np.array(list(qrels[qid].values()))
            docs = np.array(docs)
            pos = np.arange(1, docs.shape[0] + 1)
            # --- binary metrics
            bin_ids = rel_ids[grades >= binarization_point]
            hit = np.isin(docs, bin_ids)
            if hit.any():
                ranks = pos[hit]
                in_map = ranks[ranks <= self.map_at_k]
                ap[qi] = np.sum(np.arange(1, in_map.shape[0] + 1) / in_map) / bin_ids.shape[0]
                for ci, cutoff in enumerate(self.mrr_at_k):
                    if ranks[0] <= cutoff:
                        rr[ci, qi] = 1.0 / ranks[0]
                for ci, cutoff in enumerate(self.recall_at_k):
                    rec[ci, qi] = np.count_nonzero(ranks <= cutoff) / bin_ids.shape[0]
            # --- graded metric
            ghit = np.isin(docs, rel_ids)
            if ghit.any():
                ranks = pos[ghit]
                grade_of = dict(zip(rel_ids.tolist(), grades.tolist()))
                g_at_rank = np.array([grade_of[d] for d in docs[ghit].tolist()])
                ideal = np.sort(grades)[::-1]
                for ci, cutoff in enumerate(self.ndcg_at_k):
                    n_id = min(rel_ids.shape[0], cutoff)
                    idcg = np.sum(ideal[:cutoff] / np.log2(1 + np.arange(1, n_id + 1)))
                    keep = ranks <= cutoff
                    ndcg[ci, qi] = np.sum(g_at_rank[keep] / np.log2(1 + ranks[keep])) / idcg
        with np.errstate(divide="ignore", invalid="ignore"):
            mrr = rr.sum(axis=-1) / evaluated
            with_rel = (rr > 0).sum(axis=-1)
            out = {}
            for ci, cutoff in enumerate(self.mrr_at_k):
                out["MRR@" + str(cutoff)] = mrr[ci]
                out["QueriesWithRelevant@" + str(cutoff)] = with_rel[ci]
            for ci, cutoff in enumerate(self.recall_at_k):
                out["Recall@" + str(cutoff)] = rec[ci].sum() / evaluated
            for ci, cutoff in enumerate(self.ndcg_at_k):
                out["nDCG@" + str(cutoff)] = ndcg[ci].sum() / evaluated
            out["MAP@" + str(self.map_at_k)] = ap.sum() / evaluated
        out["QueriesRanked"] = evaluated
        if return_per_query:
            return out, rr, rec, ndcg, qidx_to_qid, qrels
        return out

    def _output_per_query_metrics(self, qidx_to_qid, qrels, output_path, rr, rec, ndcg):
        with open(output_path, "w") as fh:
            w = csv.writer(fh)
            w.writerow(["query"] + [f"mrr@{k}" for k in self.mrr_at_k] + [f"recall@{k}" for k in self.recall_at_k] +
                       [f"ndcg@{k}" for k in self.ndcg_at_k])
            for qi, qid in qidx_to_qid.items():
                if qid not in qrels:
                    continue
                row = [qid]
                for table in (rr, rec, ndcg):
                    row += ["{:.3f}".format(table[d][qi]) for d in range(table.shape[0])]
                w.writerow(row)


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranking_path", required=True)
    ap.add_argument("--qrels_path", required=True)
    ap.add_argument("--is_trec", action="store_true")
    args = ap.parse_args(argv)
    ev = RankingEvaluator(args.qrels_path, is_trec=args.is_trec, show_progress_bar=True)
    print(ev.compute_metrics(args.ranking_path, return_per_query=False))


if __name__ == "__main__":
    main()
