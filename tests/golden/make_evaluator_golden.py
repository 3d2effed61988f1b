#!/usr/bin/env python3
"""Golden metrics of the reference's ``evaluation/retrieval_evaluator.py`` (build container only; IMPORTS THE REFERENCE) on a
synthetic run / qrels fixture written here (our own data): a dev-style tab-separated binary qrels file and a TREC-style
space-separated graded one, runs with 2, 3 and 4 columns, queries without qrels, queries with no relevant hit, duplicates.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_evaluator_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(HERE, "evaluator_fixture")
sys.dont_write_bytecode = True


def write_fixture(rng):
    os.makedirs(FIX, exist_ok=True)
    nq, ndoc = 24, 3000
    with open(os.path.join(FIX, "qrels.dev.tsv"), "w") as fh:
        for q in range(nq):
            if q % 9 == 8:
                continue                                     # ranked query without qrels
            for p in rng.choice(ndoc, size=rng.integers(1, 4), replace=False):
                fh.write(f"{q}\t0\t{p}\t1\n")
        fh.write("9999\t0\t5\t1\n")                          # qrels for a query that is never ranked
    with open(os.path.join(FIX, "qrels.trec.txt"), "w") as fh:
        for q in range(nq):
            if q % 7 == 6:
                continue
            for p in rng.choice(ndoc, size=rng.integers(3, 40), replace=False):
                fh.write(f"{q} Q0 {p} {rng.integers(0, 4)}\n")
    for name, cols, depth in (("run2.tsv", 2, 120), ("run3.tsv", 3, 1000), ("run4.tsv", 4, 300)):
        with open(os.path.join(FIX, name), "w") as fh:
            for q in range(nq):
                # relevant documents are pulled towards the top for most queries so that the cut-offs matter
                docs = rng.permutation(ndoc)[:depth].tolist()
                if q % 5 == 0:
                    docs[3] = docs[1]                        # a duplicated pid inside a ranking
                for r, p in enumerate(docs):
                    row = [str(q), str(p)] + ([str(r + 1)] if cols >= 3 else []) + ([f"{100.0 - 0.05 * r:.4f}"] if cols == 4 else [])
                    fh.write("\t".join(row) + "\n")


def main():
    rng = np.random.default_rng(20240)
    write_fixture(rng)
    # the module file is loaded directly: the reference's evaluation/__init__.py pulls in faiss, which this image lacks
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_retrieval_evaluator", "/root/reference/evaluation/retrieval_evaluator.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    Ref = mod.RankingEvaluator
    blob = {}
    for qrels, trec in (("qrels.dev.tsv", False), ("qrels.trec.txt", True)):
        for run in ("run2.tsv", "run3.tsv", "run4.tsv"):
            ev = Ref(os.path.join(FIX, qrels), is_trec=trec)
            d, (rr, rec, nd) = ev.compute_metrics(os.path.join(FIX, run), return_per_query=True,
                                                   per_query_metrics_path=os.path.join(FIX, f"per_query.{'trec' if trec else 'dev'}.{run}.csv"))
            tag = f"{'trec' if trec else 'dev'}.{run}"
            blob[tag + ".keys"] = np.array(list(d.keys()))
            blob[tag + ".values"] = np.array([float(v) for v in d.values()])
            blob[tag + ".rr"], blob[tag + ".rec"], blob[tag + ".ndcg"] = rr, rec, nd
    ev = Ref(os.path.join(FIX, "qrels.dev.tsv"), mrr_at_k=[5, 20], ndcg_at_k=[3, 7, 50], recall_at_k=[10, 100, 500], map_at_k=100)
    d = ev.compute_metrics(os.path.join(FIX, "run3.tsv"))
    blob["custom.keys"] = np.array(list(d.keys()))
    blob["custom.values"] = np.array([float(v) for v in d.values()])
    np.savez_compressed(os.path.join(HERE, "evaluator.npz"), **blob)
    print("wrote", len(blob), "arrays")


if __name__ == "__main__":
    main()
