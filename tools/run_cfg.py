#!/usr/bin/env python3
"""Run a few training steps of one BASELINE.json config on one GPU and print samples/s (functional + perf check)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cldrd_amd.synthetic as syn
from cldrd_amd.encoder import EncoderConfig
from cldrd_amd.models import NwayDualEncoder
from cldrd_amd.trainer import NwayTrainer

CFG = {1: dict(arch="distilbert", layers=6, B=4, N=8, L=128, loss="margin_mse"),
       2: dict(arch="distilbert", layers=6, B=8, N=32, L=128, loss="kl_div"),
       3: dict(arch="distilbert", layers=6, B=4, N=200, L=128, loss="margin_mse"),
       4: dict(arch="bert", layers=12, B=4, N=64, L=256, loss="lambda_mrr")}
ap = argparse.ArgumentParser(); ap.add_argument("--cfg", type=int, default=4); ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args(); c = CFG[a.cfg]
cfg = EncoderConfig(arch=c["arch"], n_layers=c["layers"])
model = NwayDualEncoder(cfg, share_weights=False).cuda().train()
tr = NwayTrainer(model, loss=c["loss"])
kind = "teacher" if c["loss"] in ("kl_div", "margin_mse") else "mode9"
batch = syn.nway_batch(4680, c["B"], c["N"], 30, c["L"], label_kind=kind)
batch = {k: ({kk: vv.cuda() for kk, vv in v.items()} if isinstance(v, dict) else v.cuda()) for k, v in batch.items()}
for _ in range(6): out = tr.train_step(batch)      # 3 eager steps, the graph capture, two replays
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps): out = tr.train_step(batch)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
print(f"cfg{a.cfg}: {c} -> {1e3*dt:.1f} ms/step, {c['B']/dt:.1f} samples/s, loss {out[0].item():.4f}, peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
