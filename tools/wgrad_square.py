#!/usr/bin/env python3
"""The weight-gradient (TN) kernel at square and encoder shapes, tile / split variants via environment (one process per variant)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cldrd_amd import hip_ops as ops
dev = "cuda"
torch.manual_seed(0)
for T, N1, N2 in ((4096, 4096, 4096), (8192, 3072, 3072), (32768, 3072, 768), (32768, 768, 3072), (32768, 2304, 768), (32768, 768, 768)):
    dY = (torch.rand(T, N1, device=dev) * 2 - 1).bfloat16(); X = (torch.rand(T, N2, device=dev) * 2 - 1).bfloat16()
    dW = torch.empty(N1, N2, device=dev)
    ws = torch.empty(max(1, ops.wgrad_workspace_elems(T, N1, N2)), device=dev)
    for _ in range(3): ops.wgrad(dY, X, dW, T, ws)
    torch.cuda.synchronize()
    best = 1e9
    for rnd in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.wgrad(dY, X, dW, T, ws)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    print(f"wgrad T={T} N1={N1} N2={N2}: {best*1e3:.1f} us  {2.0*T*N1*N2/best/1e9:.0f} TF/s", flush=True)
