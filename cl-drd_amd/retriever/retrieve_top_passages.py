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
from .retrieval_utils import ShardedFlatIPIndex, cap_host_threads, convert_index_to_gpu, get_embeddings_from_scratch, index_retrieve, read_index


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


def write_run_file(path, query_ids, nn_doc_ids, nn_scores, nthreads: int = 0):
    """reference :90-107: one line ``qid\\tdocid\\trank\\tscore`` per hit, queries in encode order, rank 1-based, score = repr of the fp32
    value as a Python float.  A query id that occurs more than once keeps ONE entry in the reference's dict (:90-96): its lists are
    concatenated at the position of the first occurrence and the ranks run on - reproduced here.

    Arrays in (``nn_doc_ids`` int [nq, k], ``nn_scores`` float32 [nq, k], unique ids): the 6.98 M lines of a dev-set run are formatted by
    the native writer on all host cores (csrc/runfile.hip: ~0.1 s instead of ~10 s of Python f-strings; same bytes, pinned by
    tests/test_runfile.py).  Nested lists in (what ``index_retrieve(batch=128)`` returns, as in the reference): the plain loop."""
    import numpy as np
    parent = Path(path).parent
    if not os.path.exists(parent):
        os.makedirs(parent, exist_ok=True)
    qids = list(query_ids)
    if isinstance(nn_doc_ids, np.ndarray) and isinstance(nn_scores, np.ndarray) and nn_doc_ids.ndim == 2 and nn_scores.dtype == np.float32 \
            and nn_doc_ids.shape == nn_scores.shape and len(qids) == nn_doc_ids.shape[0] and len(set(qids)) == len(qids) \
            and all(isinstance(q, (int, np.integer)) for q in qids[:1]):
        from .. import _lib
        lib = _lib.load()
        q64 = np.ascontiguousarray(np.asarray(qids, dtype=np.int64))
        d64 = np.ascontiguousarray(nn_doc_ids, dtype=np.int64)
        s32 = np.ascontiguousarray(nn_scores)
        n = lib.cldrd_write_run_file(os.fsencode(str(path)), q64.ctypes.data, d64.ctypes.data, s32.ctypes.data, q64.shape[0], d64.shape[1], int(nthreads))
        if n < 0:
            raise _lib.CldrdError(f"cldrd_write_run_file: {lib.cldrd_last_error().decode()}")
        return int(n)
    qid_to_ranks = {}
    for qid, docids, scores in zip(qids, nn_doc_ids, nn_scores):
        docids = docids.tolist() if hasattr(docids, "tolist") else docids
        scores = scores.tolist() if hasattr(scores, "tolist") else scores
        qid_to_ranks.setdefault(qid, []).extend(zip(docids, scores))
    total_rank = 0
    with open(path, "w") as f:
        for qid, ranks in qid_to_ranks.items():
            f.write("".join(f"{qid}\t{docid}\t{i + 1}\t{s}\n" for i, (docid, s) in enumerate(ranks)))
            total_rank += len(ranks)
    return total_rank


def main(args):
    for tag, word in (("train", "train"), ("dev", "dev"), ("2019", "trec19"), ("2020", "trec20")):
        if tag in args.queries_path:
            assert word in args.output_path       # reference :48-59
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    cap_host_threads()
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")          # only small host objects travel (candidate lists); the search is local

    import time
    timings = {}
    t_last = [time.perf_counter()]

    def lap(name):
        torch.cuda.synchronize()
        now = time.perf_counter()
        timings[name] = timings.get(name, 0.0) + now - t_last[0]
        t_last[0] = now

    model = NwayDualEncoder(args.model_name_or_path, share_weights=args.share_weights)
    if args.resume:
        load_checkpoint_into(model, args.resume, args.is_parallel)
    model.cuda()
    lap("model_load_s")
    if args.synthetic_queries:
        loader = SyntheticSequenceDataset(args.synthetic_queries, args.max_length, seed=4242).loader()
    else:
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(args.tokenizer_name_or_path)
        dataset = SequenceDataset.create_from_seqs_file(args.queries_path, tokenizer, args.max_length, is_query=True)
        loader = torch.utils.data.DataLoader(dataset, batch_size=512, shuffle=False, num_workers=4, collate_fn=dataset.collate_fn)
    query_embs, query_ids = get_embeddings_from_scratch(model, loader, use_fp16=True, is_query=True, show_progress_bar=True)
    lap("encode_queries_s")

    path = args.index_path
    if world > 1:
        path = path.replace(".index", f".shard{rank}of{world}.index") if ".shard" not in path else path
    index = read_index(path)
    lap("index_read_s")
    index = convert_index_to_gpu(index, local_rank, False)
    lap("index_to_gpu_s")
    index = ShardedFlatIPIndex(index, rank, world)
    # the reference converts every 128-query slice to nested Python lists (retrieval_utils.py:145-146) and loops over 7 M scalars to
    # write the file; here scores / ids stay arrays from the search to the (native) run-file writer
    nn_scores, nn_doc_ids = index_retrieve(index, query_embs, args.top_k, batch=128, as_arrays=True)
    lap("search_and_merge_s")
    if rank == 0:
        total = write_run_file(args.output_path, query_ids, nn_doc_ids, nn_scores)
        lap("run_file_s")
        print(f"average ranks per query = {total / max(1, len(set(query_ids)))}")
        st = getattr(index.local, "last_stats", {})
        print("timings: " + " ".join(f"{k}={v:.3f}" for k, v in timings.items()) +
              f" | search stats: scans={st.get('scans')} rescans={st.get('rescans')} fallback={st.get('fallback_queries')}")
    main.last_timings = timings
    main.last_search_stats = dict(getattr(index.local, "last_stats", {}) or {})
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main(get_args())
