"""``retriever/retrieve_top_passages.py`` of the reference (:28-109) on MI355X: encode the queries, load the index into HBM,
search top-k in batches of 128 and write the run file ``qid\\tdocid\\trank\\tscore``.

Same flags; extra: ``--synthetic_queries N``.  With WORLD_SIZE > 1 (one process per GPU) rank r loads index shard r, every
rank encodes all queries (they are 21 MB), and rank 0 merges the per-shard lists on the host and writes the file."""
from __future__ import annotations

import argparse
import os
from pathlib import Path

import torch

from ..dataset import SequenceDataset, SyntheticSequenceDataset
from ..models.nway_dual_encoder import NwayDualEncoder
from .index_text import load_checkpoint_into
from .retrieval_utils import ShardedFlatIPIndex, convert_index_to_gpu, get_embeddings_from_scratch, index_retrieve, read_index


# names and defaults of the reference's command line (retrieve_top_passages.py:28-40) + ours
_FLAGS = {
    "resume": dict(default=""),
    "model_name_or_path": dict(default="distilbert-base-uncased"),
    "tokenizer_name_or_path": dict(default="distilbert-base-uncased"),
    "queries_path": dict(default=""),
    "index_path": dict(default=""),
    "max_length": dict(default=30),
    "top_k": dict(default=1000),
    "is_parallel": dict(default=True),
    "share_weights": dict(action="store_true", default=False),
    "output_path": dict(default=""),
    "synthetic_queries": dict(type=int, default=0),       # ours: N generated queries instead of --queries_path
}


def get_args(argv=None):
    ap = argparse.ArgumentParser(description="encode queries, search the flat index (top-k), write the run file")
    for name, kw in _FLAGS.items():
        ap.add_argument("--" + name, **kw)
    args = ap.parse_args(argv)
    args.max_length, args.top_k = int(args.max_length), int(args.top_k)
    return args


def write_run_file(path, query_ids, nn_doc_ids, nn_scores):
    """reference :90-107: queries in encode order, rank 1-based, score = repr of the fp32 value as a Python float."""
    parent = Path(path).parent
    if not os.path.exists(parent):
        os.makedirs(parent, exist_ok=True)
    total_rank = 0
    with open(path, "w") as f:
        for qid, docids, scores in zip(query_ids, nn_doc_ids, nn_scores):
            for i, (docid, s) in enumerate(zip(docids, scores)):
                f.write(f"{qid}\t{docid}\t{i + 1}\t{s}\n")
            total_rank += len(docids)
    return total_rank


def main(args):
    for tag, word in (("train", "train"), ("dev", "dev"), ("2019", "trec19"), ("2020", "trec20")):
        if tag in args.queries_path:
            assert word in args.output_path       # reference :48-59
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")          # only small host objects travel (candidate lists); the search is local

    model = NwayDualEncoder(args.model_name_or_path, share_weights=args.share_weights)
    if args.resume:
        load_checkpoint_into(model, args.resume, args.is_parallel)
    model.cuda()
    if args.synthetic_queries:
        loader = SyntheticSequenceDataset(args.synthetic_queries, args.max_length, seed=4242).loader()
    else:
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(args.tokenizer_name_or_path)
        dataset = SequenceDataset.create_from_seqs_file(args.queries_path, tokenizer, args.max_length, is_query=True)
        loader = torch.utils.data.DataLoader(dataset, batch_size=512, shuffle=False, num_workers=4, collate_fn=dataset.collate_fn)
    query_embs, query_ids = get_embeddings_from_scratch(model, loader, use_fp16=True, is_query=True, show_progress_bar=True)

    path = args.index_path
    if world > 1:
        path = path.replace(".index", f".shard{rank}of{world}.index") if ".shard" not in path else path
    index = convert_index_to_gpu(read_index(path), local_rank, False)
    index = ShardedFlatIPIndex(index, rank, world)
    nn_scores, nn_doc_ids = index_retrieve(index, query_embs, args.top_k, batch=128)
    if rank == 0:
        total = write_run_file(args.output_path, query_ids, nn_doc_ids, nn_scores)
        print(f"average ranks per query = {total / max(1, len(query_ids))}")
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main(get_args())
