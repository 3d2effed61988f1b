#!/usr/bin/env python3
"""Where the HOST time of the index encode loop goes (profiles/r06_cfg5_full.txt: the full-size job ran at 11.7 k passages/s with the GPU waiting):
cProfile of get_embeddings_from_scratch over N batches of the synthetic token cache, the loader alone, and the search statistics of the
random-init model's embeddings (status bits of the first pass).

    python tools/index_host_profile.py [rows=200000] [workers=4]
"""
import cProfile, io, os, pstats, shutil, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np


def trace(rows, workers):
    """the encode loop at scale with a progress line every 500 batches: host phase deltas + the caching allocator's state"""
    import time_cfg5_full as T
    d = "/tmp/cldrd_hostprof"
    os.makedirs(d, exist_ok=True)
    stem = d + "/s.L256.seqcache"
    T.build_token_cache(stem, rows, 256, procs=24)
    import torch
    from cldrd_amd.dataset import CachedSequenceDataset, SequenceTokenCache
    from cldrd_amd.encoder import EncoderConfig
    from cldrd_amd.models import NwayDualEncoder
    from cldrd_amd.retriever import retrieval_utils as RU
    cache = SequenceTokenCache.load(stem, {"max_length": 256})
    torch.manual_seed(0)
    model = NwayDualEncoder(EncoderConfig(arch="distilbert"), share_weights=False).cuda().eval()
    last = {"t": time.perf_counter(), "tm": {}}

    def hook(tm):
        now = time.perf_counter()
        st = torch.cuda.memory_stats()
        d_ = {k: tm[k] - last["tm"].get(k, 0.0) for k in ("load_s", "h2d_enqueue_s", "d2h_wait_s", "gather_s")}
        print(f"batch {tm['batches']:6d}: {500 * 512 / (now - last['t']):8.0f} passages/s | per batch ms: load {2 * d_['load_s']:.2f} enqueue {2 * d_['h2d_enqueue_s']:.2f} "
              f"gpu-wait {2 * d_['d2h_wait_s']:.2f} gather {2 * d_['gather_s']:.2f} | reserved {st['reserved_bytes.all.current'] / 2**30:.1f} GiB allocated "
              f"{st['allocated_bytes.all.current'] / 2**30:.1f} GiB, device mallocs {st['num_device_alloc']} frees {st['num_device_free']} retries {st['num_alloc_retries']}", flush=True)
        last["t"], last["tm"] = now, dict(tm)
    RU.PROGRESS_HOOK = hook
    t0 = time.perf_counter()
    RU.get_embeddings_from_scratch(model, CachedSequenceDataset(cache, 0, None, 512).loader(num_workers=workers, pin_memory=True), True, False)
    print(f"total: {rows / (time.perf_counter() - t0):.0f} passages/s; {RU.get_embeddings_from_scratch.last_timings}")
    shutil.rmtree(d, ignore_errors=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "trace":
        return trace(int(sys.argv[2]) if len(sys.argv) > 2 else 2500000, int(sys.argv[3]) if len(sys.argv) > 3 else 4)
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    import time_cfg5_full as T
    d = "/tmp/cldrd_hostprof"
    os.makedirs(d, exist_ok=True)
    stem = d + "/s.L256.seqcache"
    T.build_token_cache(stem, rows, 256, procs=16)
    import torch
    from cldrd_amd.dataset import CachedSequenceDataset, SequenceTokenCache
    from cldrd_amd.encoder import EncoderConfig
    from cldrd_amd.models import NwayDualEncoder
    from cldrd_amd.retriever import retrieval_utils as RU
    print(f"torch threads {torch.get_num_threads()}, cpus {os.cpu_count()}")
    cache = SequenceTokenCache.load(stem, {"max_length": 256})
    ds = CachedSequenceDataset(cache, 0, None, 512)
    for w in (0, workers):
        t0 = time.perf_counter()
        n = 0
        for b in ds.loader(num_workers=w):
            n += 1
        print(f"loader alone, {w} workers: {1e3 * (time.perf_counter() - t0) / n:.2f} ms per batch of 512 ({n} batches)")
    torch.manual_seed(0)
    model = NwayDualEncoder(EncoderConfig(arch="distilbert"), share_weights=False).cuda().eval()
    RU.get_embeddings_from_scratch(model, CachedSequenceDataset(cache, 0, 20480, 512).loader(num_workers=workers), True, False)      # warm-up
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    emb, ids = RU.get_embeddings_from_scratch(model, ds.loader(num_workers=workers), True, False)
    pr.disable()
    dt = time.perf_counter() - t0
    print(f"encode loop: {rows / dt:.0f} passages/s, {1e3 * dt / len(ds):.2f} ms per batch; host phases {RU.get_embeddings_from_scratch.last_timings}")
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
    print(s.getvalue()[:6000])
    # the search on these embeddings
    qds = __import__("cldrd_amd.dataset", fromlist=["SyntheticSequenceDataset"]).SyntheticSequenceDataset(512, 30, seed=4242)
    q, _ = RU.get_embeddings_from_scratch(model, qds.loader(), True, True)
    P = emb
    print(f"embeddings: row norm {np.linalg.norm(P, axis=1).mean():.2f}, centred row norm mean {np.linalg.norm(P - P.mean(0), axis=1).mean():.3f} "
          f"max {np.linalg.norm(P - P.mean(0), axis=1).max():.3f}; one query's scores: mean {(P @ q[0]).mean():.3f} std {(P @ q[0]).std():.4f}; "
          f"duplicate rows {rows - len(np.unique(P[:50000].view(np.uint8).reshape(50000, -1), axis=0)) if rows >= 50000 else 'n/a'} of 50000")
    idx = RU.construct_flatindex_from_embeddings(P, np.arange(rows, dtype=np.int64))
    RU.convert_index_to_gpu(idx, 0)
    t0 = time.perf_counter()
    D, I = idx.search(q, 1000)
    print(f"search of 512 queries: {time.perf_counter() - t0:.3f} s; stats {idx.last_stats}; eps-related: max centred norm {idx._max_norm:.3f}")
    shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
