#!/usr/bin/env python3
"""What parts of the training step cost on the critical path: the cfg2 step (graph replay) with parts SKIPPED (timing only, wrong
results): the query tower's forward / backward (side stream), the AdamW + norm tail.  One process, interleaved rounds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cldrd_amd.synthetic as syn
from cldrd_amd.encoder import EncoderConfig
from cldrd_amd.models import NwayDualEncoder
from cldrd_amd.trainer import NwayTrainer

dev = torch.device("cuda")
B, N, L, Lq = 8, 32, 128, 30


def build(mode):
    torch.manual_seed(0)
    model = NwayDualEncoder(EncoderConfig(arch="distilbert"), share_weights=False).to(dev).train()
    tr = NwayTrainer(model, loss="kl_div", T=1.0, learning_rate=7e-6, warmup_steps=4000, total_steps=100000)
    qe = model.query_encoder
    if mode in ("no_q", "no_q_bwd"):
        real_enc, real_bwd = qe.encode, qe.backward_from_cls
        cache = {}
        if mode == "no_q":
            def enc(*a, **k):
                if "out" not in cache:
                    cache["out"] = real_enc(*a, **k)
                return cache["out"]
            qe.encode = enc
        qe.backward_from_cls = lambda *a, **k: None
    if mode == "no_opt":
        tr._optimizer_launches = lambda *a, **k: None
    return tr


batch = syn.nway_batch(4680, B, N, Lq, L, ragged=False, label_kind="teacher")
batch = {k: ({kk: vv.to(dev) for kk, vv in v.items()} if isinstance(v, dict) else v.to(dev)) for k, v in batch.items()}
modes = sys.argv[1:] or ["full", "no_q_bwd", "no_q", "no_opt"]
trs = {m: build(m) for m in modes}
for m, tr in trs.items():
    for _ in range(8):
        tr.train_step(batch)
torch.cuda.synchronize()
res = {m: [] for m in modes}
for rnd in range(3):
    for m, tr in trs.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            tr.train_step(batch)
        torch.cuda.synchronize()
        res[m].append((time.perf_counter() - t0) / 20 * 1e3)
for m in modes:
    print(f"{m:10s} ms/step min {min(res[m]):.3f}  all {[round(x, 3) for x in res[m]]}")
