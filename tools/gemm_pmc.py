#!/usr/bin/env python3
"""A few launches of selected GEMM shapes (for rocprofv3 --pmc passes, tools/pmc_gemm.sh)."""
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
    for _ in range(4):
        ops.gemm_nt(A, B, out, bias=bias)
    torch.cuda.synchronize()
for name, N1, N2 in [("w_ffn1", 3072, 768)]:
    dY = (torch.randn(T, N1, device=dev) * 0.02).bfloat16()
    X = torch.randn(T, N2, device=dev).bfloat16()
    dW = torch.empty(N1, N2, device=dev)
    ws = torch.empty(ops.wgrad_workspace_elems(T, N1, N2), device=dev)
    for _ in range(4):
        ops.wgrad(dY, X, dW, T, ws)
    torch.cuda.synchronize()
