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


def _shard_lists(seed, world, nq, k, rows_per_shard, tie_levels=0, short=()):
    """Synthetic per-shard top-k lists in the search's output contract: scores descending, ties ordered by row, global ids = row positions,
    shards listed in `short` hold fewer than k rows (-1 / -inf padding at the tail).  `tie_levels` > 0 quantises the scores so that equal
    scores occur inside and ACROSS shards (the tie rule decides)."""
    rng = np.random.default_rng(seed)
    Ds, Is = [], []
    for r in range(world):
        have = k if r not in short else max(1, k // 3)
        rows = np.stack([np.sort(rng.choice(rows_per_shard, size=have, replace=False)) for _ in range(nq)])
        sc = rng.standard_normal((nq, have)).astype(np.float32) * 3.0 + 10.0
        if tie_levels:
            sc = np.round(sc * tie_levels) / np.float32(tie_levels)
        order = np.lexsort((rows, -sc.astype(np.float64)), axis=1)
        sc, rows = np.take_along_axis(sc, order, axis=1), np.take_along_axis(rows, order, axis=1)
        D = np.full((nq, k), -np.inf, dtype=np.float32)
        I = np.full((nq, k), -1, dtype=np.int64)
        D[:, :have], I[:, :have] = sc, rows + r * rows_per_shard
        Ds.append(D)
        Is.append(I)
    return Ds, Is


@pytest.mark.parametrize("tie_levels,short", [(0, ()), (4, ()), (4, (2, 7)), (1, (0, 1, 2, 3, 4, 5, 6))])
def test_native_merge_equals_oracle_merge_with_ties_and_padding(tie_levels, short):
    Ds, Is = _shard_lists(5, 8, 64, 100, 5000, tie_levels, short)
    for k in (100, 37):
        a = RU.merge_shard_results(Ds, Is, k)
        b = R.merge_shard_results(Ds, Is, k)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])
    # fewer candidates than k in total: (-inf, -1) padding
    a = RU.merge_shard_results(Ds[:1], Is[:1], 150)
    b = R.merge_shard_results(Ds[:1], Is[:1], 150)
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0]) and (a[1][:, 100:] == -1).all()


def test_native_merge_accepts_unsorted_lists():
    """A list that is not sorted (not what a search returns) takes the partial-sort path: same order as the oracle's full sort."""
    rng = np.random.default_rng(9)
    Ds = [rng.standard_normal((16, 50)).astype(np.float32) for _ in range(3)]
    Is = [rng.permutation(150)[:50][None, :].repeat(16, 0).astype(np.int64) + 1000 * r for r in range(3)]
    Is[1][:, 7] = -1                      # a missing entry in the middle of a list
    a = RU.merge_shard_results(Ds, Is, 60)
    b = R.merge_shard_results(Ds, Is, 60)
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])


def test_native_merge_at_cfg5_size_is_fast():
    """cfg5: 8 shards x 6 980 queries x top-1000.  Round 4's np.lexsort merge took 15.6 s; the native k-way merge
    has to stay far below the search it follows (asked: <= 30 ms on the GPU box's host; this container has 8 cores: bound 0.5 s here)."""
    import time
    Ds, Is = _shard_lists(6, 8, 6980, 1000, 1105228, tie_levels=64)
    RU.merge_shard_results([d[:8] for d in Ds], [i[:8] for i in Is], 1000)       # load the library outside the clock
    t0 = time.perf_counter()
    D, I = RU.merge_shard_results(Ds, Is, 1000)
    dt = time.perf_counter() - t0
    sel = np.arange(0, 6980, 97)
    Dr, Ir = R.merge_shard_results([d[sel] for d in Ds], [i[sel] for i in Is], 1000)
    assert np.array_equal(I[sel], Ir) and np.array_equal(D[sel], Dr)
    assert (np.diff(D.astype(np.float64), axis=1) <= 0).all()
    assert dt < 0.5, f"native merge of 8 x 6980 x 1000 took {dt:.3f} s"


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
