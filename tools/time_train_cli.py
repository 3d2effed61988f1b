#!/usr/bin/env python3
"""The TRAINING command line as a job (the analogue of tools/time_cfg5_full.py for `python -m cldrd_amd.trainer.nway_listwise`): cfg2-shaped synthetic
batches (DistilBERT x2, N = 32, per-GPU batch 8, passages of up to 128 tokens, queries 30) through `train(args)` - the loader's worker processes,
batch_to_device, train_step, logging every 50 steps - and the wall clock per step after the warm-up, next to bench.py's resident-batch number.
The host's share of a step is split three ways: waiting for the loader, batch_to_device, train_step (enqueue only: nothing here waits for the GPU
except through the queue depth).

    python tools/time_train_cli.py [steps=400] [--fixed] [--workers N] [--profile] [--ddp1]
        --fixed: every passage 128 tokens (the bench's headline batch: one shape, replayed as a HIP graph); default: MS MARCO-shaped lengths
        --ddp1:  the data-parallel code path (bucket hooks, RCCL all-reduces, agreement step) with a process group of ONE rank (CLDRD_FORCE_DDP=1)
"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ddp1 = "--ddp1" in sys.argv
if ddp1:
    os.environ.update({"CLDRD_FORCE_DDP": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": os.environ.get("MASTER_PORT", "29533"), "RANK": "0",
                       "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
import torch
from cldrd_amd.trainer import nway_listwise as T

steps = next((int(a) for a in sys.argv[1:] if a.isdigit()), 400)
fixed = "--fixed" in sys.argv
workers = int(sys.argv[sys.argv.index("--workers") + 1]) if "--workers" in sys.argv else 4
with tempfile.TemporaryDirectory(dir="/tmp") as td:
    argv = ["--experiment_folder", td, "--run_folder", "run", "--synthetic_steps", str(steps), "--synthetic_model", "distilbert", "--synthetic_nway", "32",
            "--passage_max_len", "128", "--query_max_len", "30", "--train_batch_size", "8", "--logging_steps", "50", "--evaluate_steps", "1000000",
            "--num_train_epochs", "1", "--loss", "kl_div", "--label_mode", "9", "--loader_workers", str(workers)] + (["--synthetic_fixed"] if fixed else []) \
        + (["--local_rank", "0"] if ddp1 else [])
    args = T.set_env(T.get_args(argv))
    stamps, spent = [], {"batch_to_device": 0.0, "train_step": 0.0}
    real_step, real_move = T.NwayTrainer.train_step, T.batch_to_device
    warm = 20

    def timed_step(self, batch):
        t = time.perf_counter()
        out = real_step(self, batch)
        now = time.perf_counter()
        if len(stamps) >= warm:
            spent["train_step"] += now - t
        stamps.append(now)
        return out

    def timed_move(batch, dev):
        t = time.perf_counter()
        out = real_move(batch, dev)
        if len(stamps) >= warm:
            spent["batch_to_device"] += time.perf_counter() - t
        return out
    T.NwayTrainer.train_step, T.batch_to_device = timed_step, timed_move
    t0 = time.perf_counter()
    if "--profile" in sys.argv:
        import cProfile, pstats
        pr = cProfile.Profile()
        tr = pr.runcall(T.train, args)
        pstats.Stats(pr).sort_stats("tottime").print_stats(22)
    else:
        tr = T.train(args)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    n = len(stamps) - 1 - warm
    per = (stamps[-1] - stamps[warm]) / n
    mv, st = spent["batch_to_device"] / (n + 1), spent["train_step"] / (n + 1)
    print(f"trainer CLI loop{' (data-parallel path, one rank over RCCL)' if ddp1 else ''}, {'fixed-length' if fixed else 'MSMARCO-shaped'} cfg2 batches, {workers} loader workers, {torch.get_num_threads()} torch host threads: "
          f"{steps} steps in {t1 - t0:.1f} s; steady state {1e3 * per:.3f} ms per step = {8 / per:.1f} samples/s (clock between train_step returns, steps "
          f"{warm}..{len(stamps) - 1}); host per step: batch_to_device {1e3 * mv:.2f} ms, train_step enqueue {1e3 * st:.2f} ms, loader wait + logging "
          f"{1e3 * (per - mv - st):.2f} ms; graph replay: {any(e['graph'] is not None for e in getattr(tr, '_graphs', {}).values())}; final global_step {tr.global_step}")
