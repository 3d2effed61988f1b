#!/usr/bin/env python3
"""Time the embedding + LayerNorm backward (scatter-add of the token gradients by float atomics) at the train-step shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cldrd_amd import hip_ops as ops
dev = "cuda"; M, L, d, V = 256, 128, 768, 30522
T = M * L
g = torch.Generator(device=dev).manual_seed(0)
# MS MARCO-like ids: Zipf-distributed tokens, ~40 % padding at the tail of each sequence (pad rows carry zero gradient)
ids = (torch.rand(M, L, device=dev, generator=g) ** 4 * (V - 1000)).long() + 999
lens = torch.randint(40, L + 1, (M,), device=dev, generator=g)
pad = torch.arange(L, device=dev)[None, :] >= lens[:, None]
ids[pad] = 0
dy = torch.randn(T, d, device=dev, generator=g).bfloat16()
dy.view(M, L, d)[pad] = 0
word = torch.randn(V, d, device=dev, generator=g); pos = torch.randn(512, d, device=dev, generator=g)
gamma = torch.ones(d, device=dev); mean = torch.zeros(T, device=dev); rstd = torch.ones(T, device=dev)
dword = torch.zeros_like(word); dpos = torch.zeros_like(pos); dg = torch.zeros(d, device=dev); db = torch.zeros(d, device=dev)
partial = torch.empty(ops.ln_partial_elems(T, d), device=dev)
for p_drop in (0.0, 0.1):
    def run(): ops.embed_ln_bwd(dy, ids.view(-1), word, pos, None, gamma, mean, rstd, dword, dpos, None, dg, db, partial, T, L, p_drop, 7, accumulate=False)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print(f"embed_ln_bwd T={T} d={d} dropout={p_drop}: {e0.elapsed_time(e1)/20*1e3:.1f} us")
