#!/usr/bin/env python3
"""Attention forward / backward alone at the train-step shape (256 sequences x 12 heads x L=128), HIP-event timing.
Also the driver for rocprofv3 --pmc passes (ATTN_ITERS=2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cldrd_amd import hip_ops as ops
dev = "cuda"
nseq, L, H = int(os.environ.get("NSEQ", 256)), int(os.environ.get("L", 128)), 12
iters = int(os.environ.get("ATTN_ITERS", 20))
T, d = nseq * L, H * 64
torch.manual_seed(0)
qkv = torch.randn(T, 3 * d, device=dev).bfloat16()
dctx = torch.randn(T, d, device=dev).bfloat16()
ctx = torch.empty(T, d, device=dev, dtype=torch.bfloat16)
lse = torch.empty(nseq, H, L, device=dev)
dqkv = torch.empty(T, 3 * d, device=dev, dtype=torch.bfloat16)
for p in (0.0, 0.1):
    bits = ops.attention_drop_bits(nseq, L, H, p, dev) if os.environ.get("ATTN_BITS", "1") != "0" else None
    f = lambda: ops.attention_fwd(qkv, None, ctx, lse, nseq, L, H, dropout_p=p, seed=5, drop_bits=bits)
    b = lambda: ops.attention_bwd(qkv, None, ctx, dctx, lse, dqkv, nseq, L, H, dropout_p=p, seed=5, drop_bits=bits)
    for name, fn, nbytes in (("fwd", f, T * d * 2 * 4), ("bwd", b, T * d * 2 * 9)):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / iters
        print(f"attention {name} p={p}: {t*1e3:.1f} us  ({nbytes/t/1e9:.2f} TB/s of algorithmic traffic)")
