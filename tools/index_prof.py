#!/usr/bin/env python3
"""The index (encode) leg alone, for rocprofv3 --kernel-trace --stats: N forward passes of 512 passages x L tokens through the passage tower
(retriever/index_text.py: bs = 512, max_length 256; bench.py's index legs).  usage: index_prof.py L [iters] [ragged]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cldrd_amd.synthetic as syn
from cldrd_amd.encoder import EncoderConfig
from cldrd_amd.models import NwayDualEncoder
L = int(sys.argv[1]) if len(sys.argv) > 1 else 128
it = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ragged = len(sys.argv) > 3 and sys.argv[3] == "ragged"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = NwayDualEncoder(EncoderConfig(arch="distilbert"), share_weights=False).to(dev)
model.eval()
b = syn.seq_batch(99, 512, L, ragged=ragged)["seq"]
enc = {"input_ids": b["input_ids"].to(dev), "attention_mask": b["attention_mask"].to(dev)}
if ragged:
    lens = b["attention_mask"].sum(-1)
    lg = int(lens.max())
    enc = {"input_ids": b["input_ids"][:, :lg].contiguous().to(dev), "attention_mask": b["attention_mask"][:, :lg].contiguous().to(dev), "lengths": lens}
with torch.no_grad():
    for _ in range(2):
        model.passage_embs(enc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        model.passage_embs(enc)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
F = 6 * (8 * L * 768 * 768 + 4 * L * 768 * 3072 + 4 * L * L * 768)
print(f"L={L} ragged={ragged}: {512 * it / dt:.1f} passages/s, {1e3 * dt / it:.3f} ms per batch of 512, mfma_frac {512 * it / dt * F / 2.5e15:.4f}")
