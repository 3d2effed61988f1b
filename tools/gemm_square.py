#!/usr/bin/env python3
"""The large-M NT kernel at the square shapes the CDNA guide quotes its 256^2 8-phase template on (uniform random [-1,1) operands)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cldrd_amd import hip_ops as ops
dev = "cuda"
torch.manual_seed(0)
for n in (4096, 8192):
    A = (torch.rand(n, n, device=dev) * 2 - 1).bfloat16(); B = (torch.rand(n, n, device=dev) * 2 - 1).bfloat16()
    out = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
    for v in ("0", "1"):
        os.environ["CLDRD_GEMM_PERSIST"] = v
        for _ in range(3): ops.gemm_nt(A, B, out)
        torch.cuda.synchronize()
        best = 1e9
        for rnd in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.gemm_nt(A, B, out)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        print(f"{n}^3 persist={v}: {best*1e3:.1f} us  {2.0*n**3/best/1e9:.0f} TF/s")
