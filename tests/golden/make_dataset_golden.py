#!/usr/bin/env python3
"""Golden batches of the reference's ``dataset/nway_dataset.py`` (build container only; IMPORTS THE REFERENCE).

Writes a small synthetic fixture (our own text: queries.tsv, collection.tsv, one JSON-lines training file per example
layout) under ``nway_dataset_fixture/`` and, for every label mode, the batches the reference's ``NwayDataset`` +
``collate_fn`` produce from it with the toy tokenizer of ``tests/toy_tokenizer.py`` -> ``nway_dataset.npz``.
Only data is written.  One shim: the reference imports ``ujson``, which this image lacks; ``json`` has the same
``load`` / ``loads`` for these files and is registered under that name before the import.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_dataset_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
FIX = os.path.join(HERE, "nway_dataset_fixture")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.dont_write_bytecode = True

from toy_tokenizer import WORDS, make_tokenizer  # noqa: E402

LAYOUTS = {  # file stem -> (n relT, n negatives, label modes served by it)
    "10relT_20neg": (10, 20, ["2", "3", "4", "9"]),
    "20relT_10neg": (20, 10, ["5", "10"]),
    "30relT": (30, 0, ["6"]),
    "5relT_25neg": (5, 25, ["7", "8"]),
}


def write_fixture(rng):
    os.makedirs(FIX, exist_ok=True)
    nq, npsg = 12, 90
    with open(os.path.join(FIX, "queries.tsv"), "w") as fh:
        for q in range(nq):
            words = rng.choice(WORDS, size=rng.integers(2, 9))
            fh.write(f"{1000 + q}\t{' '.join(words)}\n")
    with open(os.path.join(FIX, "collection.tsv"), "w") as fh:
        for p in range(npsg):
            words = rng.choice(WORDS + ["zzunknown"], size=rng.integers(3, 30))
            fh.write(f"{p}\t{' '.join(words)}\n")
    for stem, (n_rel, n_neg, _) in LAYOUTS.items():
        with open(os.path.join(FIX, f"train_{stem}.jsonl"), "w") as fh:
            for q in range(nq):
                pids = rng.permutation(npsg)[:n_rel + n_neg].tolist()
                hard = n_neg // 2
                fh.write(json.dumps({"qid": 1000 + q, "relT_pids": pids[:n_rel], "most_hard_pids": pids[n_rel:n_rel + hard],
                                     "semi_hard_pids": pids[n_rel + hard:]}) + "\n")
    with open(os.path.join(FIX, "train_1rel_5neg.jsonl"), "w") as fh:
        for q in range(nq):
            pids = rng.permutation(npsg)[:6].tolist()
            fh.write(json.dumps({"qid": 1000 + q, "rel_pid": pids[0], "neg_pids": pids[1:]}) + "\n")


def main():
    rng = np.random.default_rng(4680)
    write_fixture(rng)
    sys.modules["ujson"] = json
    sys.path.insert(0, "/root/reference")
    from dataset.nway_dataset import NwayDataset as RefDataset
    tok = make_tokenizer()
    q, c = os.path.join(FIX, "queries.tsv"), os.path.join(FIX, "collection.tsv")
    blob = {}

    def dump(tag, ds, idxs):
        batch = ds.collate_fn([ds[i] for i in idxs])
        for k in ("qid", "relT_pids", "neg_pids", "nway_pids"):
            blob[f"{tag}.{k}"] = np.asarray(batch[k])
        blob[f"{tag}.labels"] = batch["labels"].numpy()
        for side in ("query", "nway_passages"):
            for k in ("input_ids", "attention_mask"):
                blob[f"{tag}.{side}.{k}"] = batch[side][k].numpy()
        blob[f"{tag}.idxs"] = np.asarray(idxs)

    ctor = {"10relT_20neg": "create_from_10relT_20neg_file", "20relT_10neg": "create_from_20relT_10neg_file",
            "30relT": "create_from_30relT_file", "5relT_25neg": "create_from_5relT_25neg_file"}
    for stem, (_, _, modes) in LAYOUTS.items():
        path = os.path.join(FIX, f"train_{stem}.jsonl")
        for mode in modes:
            name = ctor[stem] if mode not in ("2", "4") else "create_from_relT_most_semi_hard_file"
            ds = getattr(RefDataset, name)(q, c, path, tok, max_query_len=6, max_passage_len=16, label_mode=mode)
            dump(f"mode{mode}", ds, [0, 5, 11])
            if mode in ("9", "8"):      # the distributed variants: line i on rank i % nranks
                for rank in (0, 1, 2):
                    dsr = getattr(RefDataset, name)(q, c, path, tok, max_query_len=6, max_passage_len=16, label_mode=mode, rank=rank,
                                                    nranks=3)
                    dump(f"mode{mode}.rank{rank}", dsr, list(range(len(dsr))))
    ds = RefDataset.create_from_json_line_file(q, c, os.path.join(FIX, "train_1rel_5neg.jsonl"), tok, max_query_len=6,
                                               max_passage_len=16, label_mode="1")
    dump("mode1", ds, [1, 2, 3, 4])
    np.savez_compressed(os.path.join(HERE, "nway_dataset.npz"), **blob)
    print("wrote", len(blob), "arrays")
    # index path: the reference's SequenceDataset (dataset/sequence_dataset.py:31-55) over the same collection, batches of 7 rows through its
    # collate_fn (tokenise + pad to the longest row of the batch), max_length 12 (some rows are truncated)
    from dataset.sequence_dataset import SequenceDataset as RefSeq
    rs = RefSeq.create_from_seqs_file(c, tok, 12, False)
    sblob = {"n": len(rs)}
    for b, lo in enumerate(range(0, len(rs), 7)):
        batch = rs.collate_fn([rs[i] for i in range(lo, min(len(rs), lo + 7))])
        sblob[f"b{b}.input_ids"] = batch["seq"]["input_ids"].numpy()
        sblob[f"b{b}.attention_mask"] = batch["seq"]["attention_mask"].numpy()
        sblob[f"b{b}.id"] = np.asarray(batch["id"])
    np.savez_compressed(os.path.join(HERE, "sequence_dataset.npz"), **sblob)
    print("wrote sequence_dataset.npz:", len(rs), "rows")


if __name__ == "__main__":
    main()
