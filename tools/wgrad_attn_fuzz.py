#!/usr/bin/env python3
"""Randomised checks against fp64 on the same rounded inputs: the grouped weight-gradient launch (random token counts incl. non-multiples of 64,
several problems of different shapes per group, with / without bias gradients) and attention forward / backward, padded and on packed rows (random nseq, L <= 256, heads,
right-padded masks incl. one-token sequences).  usage: tools/wgrad_attn_fuzz.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cldrd_amd import hip_ops as ops
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
DEV = "cuda"
bad = 0
SH = [(768, 768), (2304, 768), (3072, 768), (768, 3072), (256, 192), (128, 128), (1536, 768), (256, 384)]
for c in range(cases):
    # ---- grouped weight gradients
    n = int(rng.integers(1, 6))
    T = int(rng.choice([1, 63, 64, 65, 240, 1000, 4097, 8192, 20000]))
    q = ops.WgradQueue()
    g = torch.Generator(device=DEV).manual_seed(int(rng.integers(1 << 30)))
    refs = []
    big = bool(rng.random() < 0.5)
    for j in range(n):
        N1, N2 = SH[int(rng.integers(0, 4 if big else len(SH)))]
        Tj = T if rng.random() < 0.7 else max(1, T // 3)
        dY = (torch.randn(Tj, N1, device=DEV, generator=g) * 0.02).to(torch.bfloat16)
        X = torch.randn(Tj, N2, device=DEV, generator=g).to(torch.bfloat16)
        dW = torch.full((N1, N2), float("nan"), device=DEV)
        db = torch.full((N1,), float("nan"), device=DEV) if rng.random() < 0.6 else None
        q.add(dY, X, dW, Tj, dbias=db)
        refs.append((dY.double().T @ X.double(), dY.double().sum(0), dW, db, (Tj, N1, N2)))
    try:
        q.flush(accumulate=False)
    except Exception as e:
        print(f"case {c} wgrad group of {n}: refused ({str(e)[:100]})", flush=True)
        refs = []
    for rw, rb, dW, db, shp in refs:
        sw = rw.abs().max().item()
        ok = bool(((dW.double() - rw).abs() <= 1e-4 * rw.abs() + 2e-5 * sw + 1e-7).all()) and (db is None or bool(((db.double() - rb).abs() <= 1e-4 * rb.abs() + 2e-5 * rb.abs().max().item() + 1e-6).all()))
        if not ok:
            bad += 1
            print(f"MISMATCH case {c} wgrad {shp}: max err {(dW.double() - rw).abs().max().item():.3e} (scale {sw:.3e})", flush=True)
    # ---- attention
    nseq, L, H = int(rng.choice([1, 2, 5, 40, 300])), int(rng.choice([1, 2, 8, 31, 32, 33, 64, 100, 128, 129, 200, 256])), int(rng.choice([1, 2, 12]))
    if nseq * L * H > 400000: nseq = max(1, 400000 // (L * H))
    d, Tt = H * 64, nseq * L
    qkv = torch.randn(Tt, 3 * d, device=DEV, generator=g).to(torch.bfloat16)
    lens = rng.integers(1, L + 1, nseq); lens[0] = L
    mask = torch.from_numpy((np.arange(L)[None, :] < lens[:, None]).astype(np.int64)).to(DEV)
    x = qkv.double().view(nseq, L, 3, H, 64).requires_grad_(True)
    qq, kk, vv = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    s = (qq @ kk.transpose(2, 3) * 0.125).masked_fill(mask[:, None, None, :] == 0, -1e30)
    ref = (torch.softmax(s, -1) @ vv).transpose(1, 2).reshape(Tt, d)
    dctx = torch.randn(Tt, d, device=DEV, generator=g).to(torch.bfloat16)
    valid = mask.bool().reshape(-1)
    (ref * dctx.double() * valid[:, None]).sum().backward()
    ctx = torch.empty(Tt, d, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(nseq, H, L, dtype=torch.float32, device=DEV)
    ops.attention_fwd(qkv, mask, ctx, lse, nseq, L, H)
    okf = bool(((ctx.double() - ref.detach()).abs()[valid] <= 2.0 ** -6 * ref.detach().abs()[valid] + 2e-2).all())
    dq = torch.zeros(Tt, 3 * d, dtype=torch.bfloat16, device=DEV)
    dctx_m = (dctx.float() * valid[:, None]).to(torch.bfloat16)           # gradients of padded query rows never reach a loss
    ops.attention_bwd(qkv, mask, ctx, dctx_m, lse, dq, nseq, L, H)
    gr = x.grad.reshape(Tt, 3 * d)
    sc = gr.abs().max().item()
    okb = bool(((dq.double() - gr).abs() <= 2.0 ** -5 * gr.abs() + 2e-2 * sc + 1e-6).all())
    if not (okf and okb):
        bad += 1
        print(f"MISMATCH case {c} attention nseq {nseq} L {L} H {H}: fwd ok {okf}, bwd ok {okb} (max bwd err {(dq.double() - gr).abs().max().item():.3e}, scale {sc:.3e})", flush=True)
    # ---- the same attention on PACKED rows (cldrd_attention_*_varlen): bit for bit the padded result on the rows that exist
    cu = torch.zeros(nseq + 1, dtype=torch.int32, device=DEV)
    cu[1:] = torch.from_numpy(np.cumsum(lens)).to(DEV).to(torch.int32)
    Tp = int(cu[-1])
    tok = torch.nonzero(valid).reshape(-1)
    qkv_p, dctx_p = qkv[tok].contiguous(), dctx_m[tok].contiguous()
    ctx_p = torch.full((Tp, d), float("nan"), dtype=torch.bfloat16, device=DEV)
    lse_p = torch.full((nseq, H, L), float("nan"), dtype=torch.float32, device=DEV)
    ops.attention_fwd(qkv_p, None, ctx_p, lse_p, nseq, L, H, cu=cu)
    dq_p = torch.full((Tp, 3 * d), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.attention_bwd(qkv_p, None, ctx_p, dctx_p, lse_p, dq_p, nseq, L, H, cu=cu)
    lv = mask.bool()[:, None, :].expand(nseq, H, L)
    okp = torch.equal(ctx_p, ctx[tok]) and torch.equal(dq_p, dq[tok]) and torch.equal(lse_p[lv], lse[lv]) and bool(torch.isfinite(lse_p).all())
    if not okp:
        bad += 1
        print(f"MISMATCH case {c} packed attention nseq {nseq} L {L} H {H}: ctx {torch.equal(ctx_p, ctx[tok])} dqkv {torch.equal(dq_p, dq[tok])} "
              f"lse {torch.equal(lse_p[lv], lse[lv])}", flush=True)
print(f"{cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
