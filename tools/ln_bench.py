#!/usr/bin/env python3
"""Time the LayerNorm backward kernel (+ its partial reduction) at the train-step shape, with and without the dropout branch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cldrd_amd import hip_ops as ops, _lib
dev = "cuda"; T, d = 32768, 768
dy = torch.randn(T, d, device=dev).bfloat16(); x = torch.randn(T, d, device=dev).bfloat16()
mean = torch.zeros(T, device=dev); rstd = torch.ones(T, device=dev); gamma = torch.ones(d, device=dev)
dx = torch.empty_like(x); dx2 = torch.empty_like(x)
dg = torch.zeros(d, device=dev); db = torch.zeros(d, device=dev); dbias = torch.zeros(d, device=dev)
partial = torch.empty(_lib.load().cldrd_ln_partial_blocks(T) * 3 * d, device=dev)
for name, kw in (("plain", dict(dx_dropped=None, dropout_p=0.0)), ("dropout", dict(dx_dropped=dx2, dropout_p=0.1))):
    def run(): ops.layernorm_bwd(dy, x, mean, rstd, gamma, dx, kw["dx_dropped"], dg, db, dbias, partial, T, dropout_p=kw["dropout_p"], seed=7)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20
    nbytes = T * d * 2 * (3 if kw["dx_dropped"] is None else 4)
    print(f"ln_bwd {name}: {t*1e3:.1f} us  {nbytes/t/1e9:.2f} TB/s (kernel + partial reduction)")
