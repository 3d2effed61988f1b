#!/usr/bin/env python3
"""Index (encode) leg at different sequences-per-forward: does a forward whose FFN activations fit the 256-MB Infinity Cache
(256 sequences x 128 tokens: h = 200 MB) beat the reference's batch of 512 run as one forward?  passages/s, HIP-synchronised wall clock."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cldrd_amd.synthetic as syn
from cldrd_amd.encoder import EncoderConfig
from cldrd_amd.models import NwayDualEncoder
dev = torch.device("cuda")
cfg = EncoderConfig(arch="distilbert")
model = NwayDualEncoder(cfg, share_weights=False).to(dev).eval()
L = 128
ib = syn.seq_batch(99, 512, L)["seq"]
ids, mask = ib["input_ids"].to(dev), ib["attention_mask"].to(dev)
with torch.no_grad():
    for chunk in (512, 256, 128, 512, 256):
        f = lambda: [model.passage_embs({"input_ids": ids[i:i + chunk], "attention_mask": mask[i:i + chunk]}) for i in range(0, 512, chunk)]
        for _ in range(2): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): f()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"sequences per forward {chunk:4d}: {512 * 10 / dt:9.0f} passages/s")
