#!/usr/bin/env python3
"""Soak run: N training steps of cfg2 on a rotating set of synthetic batches (graph replay; `packed`: 64 MS MARCO-shaped batches with their token
counts - the eager, packed step whose row count changes every step), printing loss / memory / finiteness every 500 steps - nothing may grow or turn
non-finite; the loss must go down.  usage: tools/soak.py [steps] [packed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cldrd_amd.synthetic as syn
from cldrd_amd.encoder import EncoderConfig
from cldrd_amd.models import NwayDualEncoder
from cldrd_amd.trainer import NwayTrainer
steps = next((int(a) for a in sys.argv[1:] if a.isdigit()), 3000)
packed = "packed" in sys.argv
cfg = EncoderConfig(arch="distilbert", n_layers=6)
torch.manual_seed(0)
model = NwayDualEncoder(cfg, share_weights=False).cuda().train()
tr = NwayTrainer(model, loss="kl_div", learning_rate=2e-5, warmup_steps=100, total_steps=steps)
dev = lambda b: {k: ({kk: vv.cuda() for kk, vv in v.items()} if isinstance(v, dict) else v.cuda()) for k, v in b.items()}
if packed:
    from cldrd_amd.trainer.nway_listwise import batch_to_device
    batches = [batch_to_device(syn.nway_batch(100 + i, 8, 32, 30, 128, ragged=True, label_kind="teacher"), torch.device("cuda", 0)) for i in range(64)]
    assert all("lengths" in b["nway_passages"] for b in batches)
else:
    batches = [dev(syn.nway_batch(100 + i, 8, 32, 30, 128, label_kind="teacher")) for i in range(16)]
t0 = time.perf_counter()
first = None
for s in range(steps):
    out = tr.train_step(batches[s % len(batches)])
    if s % 500 == 0 or s == steps - 1:
        torch.cuda.synchronize()
        loss = float(out[0])
        first = loss if first is None else first
        print(f"step {s}: loss {loss:.4f} lr {tr.lr():.2e} finite {bool(torch.isfinite(tr.flat_p).all())} mem {torch.cuda.memory_allocated()/2**30:.2f} GiB "
              f"peak {torch.cuda.max_memory_allocated()/2**30:.2f} GiB reserved {torch.cuda.memory_reserved()/2**30:.2f} GiB  "
              f"{1e3*(time.perf_counter()-t0)/(s+1):.2f} ms/step", flush=True)
assert torch.isfinite(tr.flat_p).all().item() and loss < first, (first, loss)
assert packed or any(e["graph"] is not None for e in tr._graphs.values())
print("soak OK" + (" (packed, eager)" if packed else " (graph replay)"))
