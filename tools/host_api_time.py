#!/usr/bin/env python3
"""Wall time of the reference-shaped host search API (numpy in, numpy out) on one cfg5 shard: FlatIPIndex.search, three calls.  Usage: python tools/host_api_time.py [rows] [nq]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cldrd_amd.retriever.retrieval_utils import FlatIPIndex

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1105228
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 6980
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
P = torch.randn(rows, 768, device=dev, generator=g)
idx = FlatIPIndex.from_device_rows(P)
q = torch.randn(nq, 768, device=dev, generator=g).cpu().numpy()
if os.environ.get("PIN_PROBE"):
    for mb in (28, 56, 28, 56):
        t0 = time.perf_counter()
        h = torch.empty(mb << 18, dtype=torch.float32, pin_memory=True)
        print(f"pinned alloc of {mb} MiB: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
        del h
idx.search(q[:256], 1000)        # kernels loaded, workspaces allocated: what is timed below is the API
for i in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    D, I = idx.search(q, 1000)
    dt = time.perf_counter() - t0
    print(f"call {i}: {dt * 1e3:.1f} ms = {nq / dt:.0f} queries/s  (D {D.dtype} {D.shape}, I {I.dtype} {I.shape}, I[0,:3] {I[0,:3]})", flush=True)
    keep = (D, I) if i == 0 else keep
