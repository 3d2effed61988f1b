#!/usr/bin/env python3
"""Run a few launches of selected GEMM shapes/variants (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cldrd_amd import hip_ops as ops
dev = "cuda"
T = 32768
torch.manual_seed(0)
for name, M, N, K in [("ffn2", T, 768, 3072), ("qkv", T, 2304, 768)]:
    A = torch.randn(M, K, device=dev).bfloat16()
    B = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bias = torch.randn(N, device=dev)
    for tile in ("128", "192"):
        os.environ["CLDRD_GEMM_TILE"] = tile
        for _ in range(4):
            ops.gemm_nt(A, B, out, bias=bias)
        torch.cuda.synchronize()
