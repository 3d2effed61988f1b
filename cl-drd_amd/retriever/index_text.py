"""``retriever/index_text.py`` of the reference (:30-109) on MI355X: load a checkpoint (``module.`` prefix stripped), encode
the collection with the passage tower, build the flat inner-product index with ids, write it plus ``meta.pkl``.

Same flags; extra: ``--synthetic_rows N`` (encode N synthetic MSMARCO-shaped passages instead of ``--passages_path``) and
``--rank/--world`` style sharding through torch.distributed env vars: rank r encodes and stores the contiguous shard r
(independent units, no collective)."""
from __future__ import annotations

import argparse
import os
import pickle
from collections import OrderedDict
from pathlib import Path

import numpy as np
import torch

from ..dataset import CachedSequenceDataset, SequenceDataset, SequenceTokenCache, SyntheticSequenceDataset
from ..models.nway_dual_encoder import NwayDualEncoder
from .retrieval_utils import ShardedFlatIPIndex, cap_host_threads, construct_flatindex_from_embeddings, get_embeddings_from_scratch, write_index


# flag name -> argparse keyword arguments: names and defaults of the reference's command line (index_text.py:30-41)
_FLAGS = {
    "resume": dict(default=""),
    "model_name_or_path": dict(default="distilbert-base-uncased"),
    "tokenizer_name_or_path": dict(default="distilbert-base-uncased"),
    "passages_path": dict(default=""),
    "max_length": dict(default=256),
    "index_dir": dict(default=""),
    "is_query": dict(default=False),
    "is_parallel": dict(default=True),
    "share_weights": dict(action="store_true", default=False),
    "synthetic_rows": dict(type=int, default=0),          # ours: encode N generated MSMARCO-shaped passages
    "token_cache_dir": dict(default=""),                  # ours: tokenise the collection once (dataset.SequenceTokenCache), memory-map it afterwards
    "token_cache_stem": dict(default=""),                 # ours: memory-map an EXISTING token cache by its file stem (no collection / tokenizer opened)
    "loader_workers": dict(type=int, default=2),          # ours: DataLoader workers of the token-cache path
    "bucket_window": dict(type=int, default=16384),       # ours (token-cache path): rows are batched by LENGTH inside windows of this many rows (0: 512 consecutive rows)
}


def get_args(argv=None):
    ap = argparse.ArgumentParser(description="encode a collection with the passage tower and write the flat inner-product index")
    for name, kw in _FLAGS.items():
        ap.add_argument("--" + name, **kw)
    args = ap.parse_args(argv)
    args.max_length = int(args.max_length)      # a str when given on the command line (the reference declares no type)
    os.makedirs(args.index_dir, exist_ok=True)
    return args


def load_checkpoint_into(model, path, is_parallel=True):
    """``checkpoint["state_dict"]`` of a trainer checkpoint into ``model``; DDP's ``module.`` prefix goes when ``is_parallel``
    (reference index_text.py:63-73)."""
    # weights_only=False: a reference checkpoint carries its optimizer and a pickled LambdaLR scheduler next to the weights
    # (nway_listwise_1.py:418-426), which torch >= 2.6 refuses under the default weights_only=True
    sd = torch.load(path, map_location="cpu", weights_only=False)["state_dict"]
    if is_parallel:
        sd = OrderedDict((k[len("module."):] if k.startswith("module.") else k, v) for k, v in sd.items())
    model.load_state_dict(sd)


def collection_loader(path, tokenizer, max_length, is_query, token_cache_dir, rank, world, bucket_window=0):
    """Batches of 512 rows of this rank's contiguous row range (reference index_text.py:84: bs 512, 4 workers).  With a token cache
    the rank memory-maps the tokenised collection and reads only its own rows; without, it parses the TSV and tokenises per batch as the
    reference does."""
    if token_cache_dir:
        cache = SequenceTokenCache.open_or_build(token_cache_dir, path, tokenizer, max_length, rank, world)
        lo, hi = ShardedFlatIPIndex.shard_bounds(len(cache), world, rank)
        return CachedSequenceDataset(cache, lo, hi, batch_size=512, pad_id=int(getattr(tokenizer, "pad_token_id", 0) or 0),
                                     bucket_window=bucket_window).loader(pin_memory=True)
    dataset = SequenceDataset.create_from_seqs_file(path, tokenizer, max_length, is_query=is_query)
    lo, hi = ShardedFlatIPIndex.shard_bounds(len(dataset), world, rank)
    dataset.ids, dataset.seqs = dataset.ids[lo:hi], dataset.seqs[lo:hi]
    return torch.utils.data.DataLoader(dataset, batch_size=512, shuffle=False, num_workers=4, collate_fn=dataset.collate_fn)


def main(args):
    import time
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    cap_host_threads()
    timings = {}
    t_last = [time.perf_counter()]

    def lap(name):
        torch.cuda.synchronize()
        now = time.perf_counter()
        timings[name] = timings.get(name, 0.0) + now - t_last[0]
        t_last[0] = now
    model = NwayDualEncoder(args.model_name_or_path, share_weights=args.share_weights)
    print("************************* share weights = {} *************************".format(args.share_weights))
    if args.resume:
        print(f"load model from ==> {args.resume}")
        load_checkpoint_into(model, args.resume, args.is_parallel)
    model.cuda()
    lap("model_load_s")

    if getattr(args, "token_cache_stem", ""):
        cache = SequenceTokenCache.load(args.token_cache_stem, {"max_length": args.max_length})
        lo, hi = ShardedFlatIPIndex.shard_bounds(len(cache), world, rank)
        text_loader = CachedSequenceDataset(cache, lo, hi, batch_size=512, bucket_window=int(getattr(args, "bucket_window", 0))).loader(
            num_workers=int(getattr(args, "loader_workers", 2)), pin_memory=True)
    elif args.synthetic_rows:
        lo, hi = ShardedFlatIPIndex.shard_bounds(args.synthetic_rows, world, rank)
        dataset = SyntheticSequenceDataset(hi - lo, args.max_length, first_id=lo)
        text_loader = dataset.loader()
    else:
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(args.tokenizer_name_or_path)
        text_loader = collection_loader(args.passages_path, tokenizer, args.max_length, args.is_query, args.token_cache_dir, rank, world,
                                        bucket_window=int(getattr(args, "bucket_window", 0)))

    lap("open_collection_s")
    text_embs, text_ids = get_embeddings_from_scratch(model, text_loader, use_fp16=True, is_query=args.is_query, show_progress_bar=True)
    lap("encode_s")
    text_id_to_idx = {tid: idx for idx, tid in enumerate(text_ids)}
    print("embs dtype: ", text_embs.dtype)
    index = construct_flatindex_from_embeddings(text_embs, np.array(text_ids))
    lap("index_build_s")
    stem = Path(args.resume).stem.split(".")[0] if args.resume else "random_init"
    index_path = os.path.join(args.index_dir, stem + (f".shard{rank}of{world}" if world > 1 else "") + ".index")
    write_index(index, index_path)
    lap("index_write_s")
    with open(os.path.join(args.index_dir, "meta.pkl" if world == 1 else f"meta.shard{rank}.pkl"), "wb") as f:
        pickle.dump({"text_ids": np.array(text_ids), "text_id_to_idx": text_id_to_idx}, f)
    lap("meta_pkl_s")
    timings["encode_host_phases"] = dict(getattr(get_embeddings_from_scratch, "last_timings", {}))
    main.last_timings = timings
    print("timings: " + " ".join(f"{k}={v:.3f}" for k, v in timings.items() if not isinstance(v, dict)))
    return index_path


if __name__ == "__main__":
    main(get_args())
