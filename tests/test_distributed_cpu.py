"""N>1 paths on CPU with gloo, world_size 2: the gradient-bucket all-reduce of the trainer and the
sharded retrieve (gather + host merge).  The kernels are GPU-only, so the local search is stood in for by the CPU oracle:
what is under test here is the partitioning, the exchange and the merge."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cldrd_amd.synthetic as syn
from cldrd_amd.retriever import retrieval_utils as RU
from cldrd_amd.trainer import nway_listwise as TL
from oracle import retrieval_ref as R


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)


class _OracleLocal:
    """CPU stand-in for FlatIPIndex (same search contract)."""
    def __init__(self, emb, ids):
        self.emb, self.ids, self.ntotal = emb, ids, emb.shape[0]

    def search(self, q, k):
        return R.flat_ip_search(self.emb, self.ids, q, k)


def _retrieve_worker(rank, world, port, out):
    _init(rank, world, port)
    emb = syn.corpus_embeddings(21, 4001, 32)
    q = syn.corpus_embeddings(22, 6, 32)
    lo, hi = RU.ShardedFlatIPIndex.shard_bounds(4001, world, rank)
    idx = RU.ShardedFlatIPIndex(_OracleLocal(emb[lo:hi], np.arange(lo, hi, dtype=np.int64)), rank, world)
    D, I = idx.search(q, 25)
    if rank == 0:
        Dr, Ir = R.flat_ip_search(emb, None, q, 25)
        assert np.array_equal(I, Ir) and np.array_equal(D, Dr)
        open(out, "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_retrieve_gloo(tmp_path, world):
    """world 8 = the cfg5 layout: rank 0's merge of 8 row-range shards equals the global search."""
    out = str(tmp_path / "ok")
    mp.spawn(_retrieve_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert open(out).read() == "ok"


def _allreduce_worker(rank, world, port, out):
    _init(rank, world, port)
    n = 64 * 40
    buckets = [(0, 0, 0, 640), (0, 1, 640, 1600), (0, -1, 1600, n)]
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    TL.allreduce_buckets(g, buckets, world)
    expect = torch.arange(n, dtype=torch.float32) * float(world * (world + 1) // 2)     # (1 + 2 + ..) * x: SUM over ranks (the mean is folded into dlogits)
    assert torch.equal(g, expect)
    # rank-sharded data: the reference's line_idx % nranks == rank rule (dataset/nway_dataset.py:305)
    mine = [i for i in range(10) if TL.owns_example(i, rank, world)]
    assert mine == list(range(rank, 10, world))
    if rank == 0:
        open(out, "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_gradient_bucket_allreduce_gloo(tmp_path, world):
    out = str(tmp_path / "ok")
    mp.spawn(_allreduce_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert open(out).read() == "ok"


def test_shard_bounds_cover_everything():
    for n, w in ((8841823, 8), (10, 3), (5, 8)):
        spans = [RU.ShardedFlatIPIndex.shard_bounds(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
    assert RU.ShardedFlatIPIndex.shard_bounds(8841823, 8, 0) == (0, 1105228)       # shards of 1 105 228 rows


def test_merge_matches_oracle_merge():
    emb = syn.corpus_embeddings(23, 900, 16)
    q = syn.corpus_embeddings(24, 4, 16)
    parts = [R.flat_ip_search(emb[lo:lo + 300], np.arange(lo, lo + 300), q, 12) for lo in range(0, 900, 300)]
    a = RU.merge_shard_results([p[0] for p in parts], [p[1] for p in parts], 12)
    b = R.merge_shard_results([p[0] for p in parts], [p[1] for p in parts], 12)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_run_file_writer_matches_reference_format(tmp_path):
    from cldrd_amd.retriever.retrieve_top_passages import write_run_file
    p = tmp_path / "dev" / "x.run"
    write_run_file(str(p), [11, 12], [[5, 6], [7, 8]], [[2.5, 1.25], [0.5, 0.25]])
    assert open(p).read() == "".join(R.run_file_lines([11, 12], [[5, 6], [7, 8]], [[2.5, 1.25], [0.5, 0.25]]))


def _steps_worker(rank, world, port, out):
    _init(rank, world, port)
    # line_idx % nranks sharding + drop_last leaves rank r with 7 + r batches: everybody must run MIN = 7 (reference hazard,
    # dataset/nway_dataset.py:305; a rank alone in a bucket all-reduce at the end of the epoch hangs)
    assert TL.common_steps_per_epoch(7 + rank, True) == 7
    if rank == 0:
        open(out, "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_equal_steps_per_epoch_gloo(tmp_path, world):
    out = str(tmp_path / "ok")
    mp.spawn(_steps_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert open(out).read() == "ok"


def test_bench_refuses_more_gpus_than_the_node_has():
    """`python bench.py --gpus N` on a node with fewer than N GPUs (this container: none; the GPU box: one) must end at once with a
    clear message - before any rank is spawned and before any rendezvous a missing rank would leave hanging."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = torch.cuda.device_count() + 1
    for env_extra in ({}, {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29555"}):
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
        env.update(env_extra)
        t0 = time.time()
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                           env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and "visible GPUs" in (r.stderr + r.stdout), (r.returncode, r.stderr[-500:])
        assert time.time() - t0 < 60
