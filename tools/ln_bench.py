#!/usr/bin/env python3
"""Time the LayerNorm backward kernel at the train-step shape (T = 32768, d = 768) in the formats of the all-fp16 mode: fp16 gradient stream
(+ fp16 branch), fp32 pre-LN sums; with / without dropout; operands rotated over ROT buffer sets so that nothing is served from the Infinity Cache
(as in the step, where the pre-LN sums were written ~8 ms earlier)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cldrd_amd import hip_ops as ops, _lib
dev = "cuda"; T, d = 32768, 768
ROT = int(os.environ.get("ROT", "6"))
stream = os.environ.get("STREAM", "fp16")
sd = torch.float16 if stream == "fp16" else torch.float32
sets = []
for i in range(ROT):
    sets.append(dict(dy=torch.randn(T, d, device=dev).to(sd), br=torch.randn(T, d, device=dev).half(), x=torch.randn(T, d, device=dev),
                     dx=torch.empty(T, d, device=dev, dtype=sd), dx2=torch.empty(T, d, device=dev, dtype=torch.float16)))
mean = torch.zeros(T, device=dev); rstd = torch.ones(T, device=dev); gamma = torch.ones(d, device=dev)
dg = torch.zeros(d, device=dev); db = torch.zeros(d, device=dev); dbias = torch.zeros(d, device=dev)
partial = torch.empty(_lib.load().cldrd_ln_partial_blocks(T) * 3 * d, device=dev)
q = ops.LnReduceQueue()
for name, p in (("plain", 0.0), ("dropout", 0.1)):
    def run(i):
        s = sets[i % ROT]
        d2 = s["dx2"] if (p > 0 or stream != "fp16") else None
        ops.layernorm_bwd(s["dy"], s["x"], mean, rstd, gamma, s["dx"], d2, dg, db, dbias, partial, T, dropout_p=p, seed=7, dy_branch=s["br"], defer=q)
        q.jobs.clear()
    for i in range(ROT): run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    N = 5 * ROT
    for i in range(N): run(i)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / N
    es = 2 if stream == "fp16" else 4
    nbytes = T * d * (es + 2 + 4 + es + (2 if (p > 0 or stream != "fp16") else 0))
    print(f"ln_bwd {stream} stream, {name}: {t*1e3:.1f} us  {nbytes/1e6:.0f} MB  {nbytes/t/1e9:.2f} TB/s")
