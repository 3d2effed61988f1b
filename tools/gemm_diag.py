#!/usr/bin/env python3
"""Diagnostics for the ring GEMM: time vs K (slope = marginal cost per K tile), vs M (CU occupancy), zero vs random data."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401  (development build: the knobs below do not exist in the product library)
import torch
from cldrd_amd import hip_ops as ops
dev = "cuda"

def run(M, N, K, tile, zero=False, iters=10):
    os.environ["CLDRD_GEMM_TILE"] = tile
    A = torch.zeros(M, K, device=dev, dtype=torch.bfloat16) if zero else torch.randn(M, K, device=dev).bfloat16()
    B = torch.zeros(N, K, device=dev, dtype=torch.bfloat16) if zero else (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    best = 1e9
    for _ in range(3):
        for _ in range(2): ops.gemm_nt(A, B, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): ops.gemm_nt(A, B, out)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best * 1e3

for tile in ("192", "128"):
    print(f"--- tile {tile}: time vs K at M=32768 N=768 (plain epilogue)")
    for K in (256, 768, 1536, 3072, 6144):
        t = run(32768, 768, K, tile)
        print(f"K={K:5d}: {t:8.1f} us  {2.0*32768*768*K/t/1e6:7.1f} TF/s", flush=True)
    print(f"--- tile {tile}: time vs M at N=768 K=3072")
    for M in (4096, 8192, 16384, 32768, 65536):
        t = run(M, 768, 3072, tile)
        print(f"M={M:6d}: {t:8.1f} us  {2.0*M*768*3072/t/1e6:7.1f} TF/s", flush=True)
    t = run(32768, 768, 3072, tile, zero=True)
    print(f"zero data M=32768 K=3072: {t:8.1f} us {2.0*32768*768*3072/t/1e6:7.1f} TF/s")
    print(f"--- tile {tile}: N sweep at M=32768 K=768")
    for N in (768, 1536, 2304, 3072):
        t = run(32768, N, 768, tile)
        print(f"N={N:5d}: {t:8.1f} us  {2.0*32768*N*768/t/1e6:7.1f} TF/s", flush=True)
