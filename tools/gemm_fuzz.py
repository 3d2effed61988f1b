#!/usr/bin/env python3
"""Randomised check of cldrd_gemm_nt16* against fp64 on the same rounded inputs: random M (1 .. 6000: both kernels, tails), N (multiples of 8),
K (multiples of 64), operand format (bf16 / fp16), and every epilogue the encoder uses (bias, GELU with / without the tape, GELU gradient,
bf16 / fp32 residual, LayerNorm on the fly, dropout, fp32 out, fp16 -> bf16 result, bf16 tape copy).  usage: tools/gemm_fuzz.py [cases] [seed]"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cldrd_amd import hip_ops as ops
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
DEV = "cuda"
bad = 0
def gelu(x): return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))
def dgelu(x): return 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2.0 * math.pi)
for c in range(cases):
    M = int(rng.choice([1, 7, 100, 240, 256, 300, 1000, 1023, 1024, 1025, 2000, 4096, 4100, 6000]))
    N = int(rng.choice([8, 64, 128, 192, 200, 256, 384, 768, 1024, 2304, 3072]))
    K = int(rng.choice([64, 128, 192, 768, 1024, 2304, 3072]))
    f16 = bool(rng.random() < 0.4)
    dt = torch.float16 if f16 else torch.bfloat16
    g = torch.Generator(device=DEV).manual_seed(int(rng.integers(1 << 30)))
    A = torch.randn(M, K, device=DEV, generator=g).to(dt)
    B = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(dt)
    ref = A.double() @ B.double().T
    flav = str(rng.choice(["plain", "bias", "gelu", "gelu_tape", "gelugrad", "res16", "res32", "res32_ln", "f32", "res32_nobias"]))
    if f16 and flav in ("gelugrad", "res16", "plain", "f32", "res32_nobias"):
        flav = "bias"
    kw = {}
    out_dt = dt
    bias = torch.randn(N, device=DEV, generator=g)
    tol_r, tol_a = 2.0 ** -7 if not f16 else 2.0 ** -9, 2e-2
    if flav == "bias":
        kw["bias"] = bias; ref = ref + bias.double()
        if f16 and rng.random() < 0.5: out_dt = torch.bfloat16; tol_r = 2.0 ** -7       # fp16 operands, bf16 result
    elif flav == "gelu":
        kw["bias"] = bias; kw["act"] = 1; ref = gelu(ref + bias.double())
    elif flav == "gelu_tape":
        kw["bias"] = bias; kw["act"] = 3; pre = torch.empty(M, N, dtype=torch.bfloat16, device=DEV); kw["preact"] = pre
        if f16: kw["out_copy"] = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        x = ref + bias.double(); ref = gelu(x); ref_pre = dgelu(x)
    elif flav == "gelugrad":
        gp = torch.randn(M, N, device=DEV, generator=g).to(torch.bfloat16); kw["gelu_pre"] = gp; kw["act"] = 2; ref = ref * gp.double()
    elif flav == "res16":
        r = torch.randn(M, N, device=DEV, generator=g).to(torch.bfloat16); kw["residual"] = r; ref = ref + r.double()
    elif flav in ("res32", "res32_ln", "res32_nobias"):
        r = torch.randn(M, N, device=DEV, generator=g); kw["residual"] = r; out_dt = torch.float32; tol_r, tol_a = 1e-4, 1e-3
        if flav != "res32_nobias": kw["bias"] = bias; ref = ref + bias.double()
        if flav == "res32_ln":
            mean, rstd = r.mean(1), 1.0 / torch.sqrt(r.var(1, unbiased=False) + 1e-12)
            gam, bet = 1 + 0.1 * torch.randn(N, device=DEV, generator=g), 0.1 * torch.randn(N, device=DEV, generator=g)
            kw["residual_ln"] = (mean.contiguous(), rstd.contiguous(), gam, bet)
            ref = ref + ((r.double() - mean.double()[:, None]) * rstd.double()[:, None] * gam.double() + bet.double())
        else:
            ref = ref + r.double()
    elif flav == "f32":
        out_dt = torch.float32; tol_r, tol_a = 1e-4, 1e-3
    out = torch.full((M, N), float("nan"), dtype=out_dt, device=DEV)
    try:
        ops.gemm_nt(A, B, out, M, **kw)
    except Exception as e:
        print(f"case {c}: M {M} N {N} K {K} f16 {f16} {flav}: refused ({str(e)[:80]})", flush=True)
        continue
    scale = ref.abs().max().item()
    err = (out.double() - ref).abs()
    ok = bool((err <= tol_r * ref.abs() + tol_a * max(scale, 1e-3) * (1 if out_dt != torch.float32 else 0.01)).all()) and bool(torch.isfinite(out.float()).all())
    if ok and flav == "gelu_tape":
        ok = bool(((pre.double() - ref_pre).abs() <= 2.0 ** -6 * ref_pre.abs() + 2e-2).all())
        if ok and "out_copy" in kw:
            ok = bool(torch.equal(kw["out_copy"].float(), out.float().to(torch.bfloat16).float())) or bool(((kw["out_copy"].double() - ref).abs() <= 2.0 ** -7 * ref.abs() + 2e-2 * scale).all())
    if not ok:
        bad += 1
        print(f"MISMATCH case {c}: M {M} N {N} K {K} f16 {f16} {flav} out {out_dt}: max err {err.max().item():.3e} (scale {scale:.3e})", flush=True)
print(f"{cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
