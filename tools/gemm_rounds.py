#!/usr/bin/env python3
"""How the large-M NT GEMM's time splits into per-launch, per-tile and per-K-tile parts: time vs M (= rounds of tiles per CU) at the encoder's
N / K, plain epilogue, ring vs persistent form.  A per-tile cost that does not shrink with more rounds is inside the tile (prologue latency,
epilogue); one that shrinks is the synchronised store / first-load burst of a round (workgroups drift apart over rounds)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cldrd_amd import hip_ops as ops

dev = "cuda"
torch.manual_seed(0)
for N, K in ((768, 768), (768, 3072), (3072, 768), (2304, 768)):
    B = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    for M in (8192, 16384, 32768, 65536, 131072):
        A = torch.randn(M, K, device=dev).bfloat16()
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        line = f"N={N:5d} K={K:5d} M={M:7d}:"
        for v in ("ring", "pers"):
            os.environ["CLDRD_GEMM_PERSIST"] = "1" if v == "pers" else "0"
            best = 1e9
            for rnd in range(3):
                for _ in range(3): ops.gemm_nt(A, B, out)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): ops.gemm_nt(A, B, out)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10)
            bn = 192 if N == 768 else 256
            tiles = (M // 256) * (N // bn)
            line += f"  {v}: {best*1e3:7.1f} us {2.0*M*N*K/best/1e9:7.1f} TF/s ({tiles/256:.2f} rounds, {best*1e3/max(1.0, tiles/256):6.1f} us/round)"
        print(line, flush=True)
        del A, out
os.environ.pop("CLDRD_GEMM_PERSIST", None)
