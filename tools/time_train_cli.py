#!/usr/bin/env python3
"""The TRAINING command line as a job (the analogue of tools/time_cfg5_full.py for `python -m cldrd_amd.trainer.nway_listwise`): cfg2-shaped synthetic
batches (DistilBERT x2, N = 32, per-GPU batch 8, passages of up to 128 tokens, queries 30) through `train(args)` - the loader, batch_to_device,
train_step (graph replay), logging every 50 steps - and the wall clock per step after the warm-up, next to bench.py's resident-batch number.

    python tools/time_train_cli.py [steps=400] [--fixed]        (--fixed: every passage 128 tokens, as the bench's headline batch)
"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cldrd_amd.trainer import nway_listwise as T

steps = next((int(a) for a in sys.argv[1:] if a.isdigit()), 400)
fixed = "--fixed" in sys.argv
with tempfile.TemporaryDirectory(dir="/tmp") as td:
    argv = ["--experiment_folder", td, "--run_folder", "run", "--synthetic_steps", str(steps), "--synthetic_model", "distilbert", "--synthetic_nway", "32",
            "--passage_max_len", "128", "--query_max_len", "30", "--train_batch_size", "8", "--logging_steps", "50", "--evaluate_steps", "1000000",
            "--num_train_epochs", "1", "--loss", "kl_div", "--label_mode", "9"]
    args = T.set_env(T.get_args(argv))
    stamps = []
    real = T.NwayTrainer.train_step

    def timed(self, batch):
        out = real(self, batch)
        stamps.append(time.perf_counter())
        return out
    T.NwayTrainer.train_step = timed
    if fixed:
        loader_cls = T._SyntheticLoader
        real_iter = loader_cls.__iter__

        def it(self):
            from cldrd_amd import synthetic as syn
            a = self.args
            for i in range(a.synthetic_steps):
                yield syn.nway_batch(a.seed + i, a.train_batch_size, a.synthetic_nway, a.query_max_len, a.passage_max_len, ragged=False, label_kind="teacher")
        loader_cls.__iter__ = it
    t0 = time.perf_counter()
    tr = T.train(args)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    warm = 20
    per = (stamps[-1] - stamps[warm]) / (len(stamps) - 1 - warm)
    print(f"trainer CLI loop, {'fixed-length' if fixed else 'MSMARCO-shaped'} cfg2 batches: {steps} steps in {t1 - t0:.1f} s; steady state {1e3 * per:.3f} ms per step = "
          f"{8 / per:.1f} samples/s (host enqueue-side clock between train_step returns, steps {warm}..{len(stamps) - 1}); final global_step {tr.global_step}")
