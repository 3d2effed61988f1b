"""Kernel-level parity on the MI355X (run with -m gpu): every C-ABI entry point against a plain fp32/fp64
PyTorch-CPU / numpy statement of the same op on the same (bf16-rounded) inputs.  Tolerances are written at
each check: bf16 outputs carry one 2^-9 relative rounding, fp32 outputs only accumulation-order noise."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import cldrd_amd.synthetic as syn
from cldrd_amd import hip_ops as ops

DEV = "cuda"


def rnd(seed, shape, scale=1.0):
    return torch.from_numpy((syn.normal(seed, int(np.prod(shape))) * scale).astype(np.float32).reshape(shape))


def bf(t):
    return t.to(torch.bfloat16)


def close(got, ref, rtol, atol, what=""):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = int((err > tol).sum())
    assert bad == 0, f"{what}: {bad}/{err.numel()} off, max err {err.max().item():.4e} (ref max {ref.abs().max().item():.3e})"


# ------------------------------------------------------------------------------------------------ GEMM NT
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 256, 128), (1000, 768, 768), (2048, 2304, 768), (240, 768, 3072),
                                   (4096, 3072, 768), (77, 384, 192),
                                   # large-M ring kernel: BN=192 / BN=256 variants, M tails, 1..4 K slices, K=3072
                                   (1100, 384, 192), (1030, 512, 64), (1024, 1024, 128), (2000, 768, 3072), (1500, 256, 320),
                                   # grouped N-tile order (K <= 1024, more N tiles than a group): M panels not a multiple of 8,
                                   # a partial last group, BN = 192 and BN = 256
                                   (1100, 3072, 768), (2900, 2304, 1024), (2300, 1920, 512), (1281, 2560, 1024)])
def test_gemm_nt_plain(M, N, K):
    A, B = bf(rnd(1, (M, K))), bf(rnd(2, (N, K)))
    ref = A.float() @ B.float().T
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt(A.to(DEV), B.to(DEV), out)
    # bf16 store: rel 2^-8; fp32 accumulation order: ~1e-6 * sqrt(K) * |a||b|
    close(out, ref, 1.0 / 128, 1e-3 * math.sqrt(K), f"gemm_nt {M}x{N}x{K}")
    out32 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm_nt(A.to(DEV), B.to(DEV), out32)
    close(out32, ref, 1e-4, 1e-4 * math.sqrt(K), f"gemm_nt f32 {M}x{N}x{K}")


def test_gemm_nt_asymmetric_identity():
    """A = I with an asymmetric B catches a transposed C write (cdna_hip_programming.md section 3)."""
    M = N = K = 128
    A = torch.eye(M)
    B = torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251 - 100.0
    out = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm_nt(bf(A).to(DEV), bf(B).to(DEV), out)
    assert torch.equal(out.cpu(), bf(B).float().T.contiguous())


@pytest.mark.parametrize("M,N,K", [(520, 384, 256), (1300, 384, 256), (1300, 512, 256)])
def test_gemm_nt_epilogues(M, N, K):
    A, B = bf(rnd(3, (M, K), 0.5)), bf(rnd(4, (N, K), 0.5))
    bias = rnd(5, (N,))
    res = bf(rnd(6, (M, N)))
    base = A.float() @ B.float().T + bias
    Ad, Bd = A.to(DEV), B.to(DEV)
    # bias + GELU with the pre-activation saved
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    pre = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt(Ad, Bd, out, bias=bias.to(DEV), preact=pre, act=1)
    close(pre, base, 1 / 128, 2e-2, "preact")
    close(out, torch.nn.functional.gelu(base), 1 / 128, 2e-2, "gelu")
    # bias + residual
    ops.gemm_nt(Ad, Bd, out, bias=bias.to(DEV), residual=res.to(DEV))
    close(out, base + res.float(), 1 / 128, 2e-2, "residual")
    # data gradient through GELU: acc * gelu'(pre) + residual, alpha
    gp = bf(rnd(7, (M, N)))
    x = gp.float()
    gelu_grad = 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)
    ops.gemm_nt(Ad, Bd, out, gelu_pre=gp.to(DEV), residual=res.to(DEV), alpha=0.5)
    close(out, 0.5 * (A.float() @ B.float().T) * gelu_grad + res.float(), 1 / 128, 2e-2, "gelu_grad")
    # derivative form (what the encoder's tape uses): the forward saves gelu'(pre-activation), the backward multiplies by it
    dsave = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt(Ad, Bd, out, bias=bias.to(DEV), preact=dsave, act=3)
    close(out, torch.nn.functional.gelu(base), 1 / 128, 2e-2, "gelu (derivative form)")
    close(dsave, 0.5 * (1 + torch.erf(base / math.sqrt(2))) + base * torch.exp(-0.5 * base * base) / math.sqrt(2 * math.pi), 1 / 128, 1e-3,
          "saved gelu'")
    ops.gemm_nt(Ad, Bd, out, gelu_pre=dsave, act=2)
    close(out, (A.float() @ B.float().T) * dsave.float().cpu(), 1 / 128, 2e-2, "multiply by the saved derivative")
    # fp32 residual given as a LayerNorm still to be applied (the fp32 residual stream without a stored LayerNorm output)
    s32 = rnd(8, (M, N), 2.0) + 0.5
    gam, bet = 1 + rnd(9, (N,), 0.1), rnd(10, (N,), 0.1)
    mu = s32.mean(1)
    rs = 1.0 / torch.sqrt(s32.var(1, unbiased=False) + 1e-12)
    ln = (s32 - mu[:, None]) * rs[:, None] * gam + bet
    out32 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm_nt(Ad, Bd, out32, bias=bias.to(DEV), residual=s32.to(DEV), residual_ln=(mu.to(DEV), rs.to(DEV), gam.to(DEV), bet.to(DEV)))
    close(out32, base + ln, 1e-5, 1e-4 * math.sqrt(K), "residual = LayerNorm(pre-LN sum)")
    ops.gemm_nt(Ad, Bd, out32, bias=bias.to(DEV), residual=s32.to(DEV), residual_ln=(mu.to(DEV), rs.to(DEV), gam.to(DEV), bet.to(DEV)),
                dropout_p=0.1, seed=77)
    import oracle.dropout_ref as DRo
    keep = torch.from_numpy(DRo.keep_mask(77, 0.1, M, N))
    close(out32, torch.where(keep, base / 0.9, torch.zeros(())) + ln, 1e-5, 1e-4 * math.sqrt(K), "dropout + LayerNorm residual")


@pytest.mark.parametrize("M,N,K", [(256, 768, 768), (256, 3072, 768), (100, 520, 832), (1, 8, 64), (640, 1600, 1024), (257, 2304, 768)])
@pytest.mark.parametrize("flavour", ["plain", "bias_gelu_dpre16", "gelugrad_d16", "bias_drop_res32_f32", "ln_res32_f32", "bf16_res16"])
def test_gemm_nt_small_m_one_launch_kernel_is_bit_identical(flavour, M, N, K):
    """Small-M problems with K <= 1024 take the 64 x 64 one-launch kernel (seven K tiles in flight): same MFMA order along K and the same
    epilogue as the 128 x 128 kernel in its one-pass form (ops.set_tuning("gemm_splitk", 1)), so every output - C, the GELU tape, ragged edges -
    is bit-identical; rows / columns outside [M, N] are not written."""
    bf = flavour.startswith("bf16")
    dt = torch.bfloat16 if bf else torch.float16
    g = torch.Generator(device=DEV).manual_seed(M * 7 + N + K)
    A = (torch.randn(M + 3, K, device=DEV, generator=g) * 0.5).to(dt)
    B = (torch.randn(N, K, device=DEV, generator=g) * 0.5).to(dt)
    bias = torch.randn(N, device=DEV, generator=g)
    kw, out_dtype = {}, dt
    if flavour == "bias_gelu_dpre16":
        kw.update(bias=bias, preact=True, act=3)
    if flavour == "gelugrad_d16":
        kw.update(gelu_pre=torch.rand(M + 3, N, device=DEV, generator=g).half(), act=2)
    if flavour == "bf16_res16":
        kw.update(bias=bias, residual=torch.randn(M + 3, N, device=DEV, generator=g).bfloat16())
    if "res32" in flavour:
        kw.update(bias=bias, residual=torch.randn(M + 3, N, device=DEV, generator=g))
        out_dtype = torch.float32
    if flavour == "bias_drop_res32_f32":
        kw.update(dropout_p=0.1, seed=(5 << 32) | 9)
    if flavour == "ln_res32_f32":
        s32 = kw["residual"]
        kw["residual_ln"] = (s32.mean(1).contiguous(), (1.0 / torch.sqrt(s32.var(1, unbiased=False) + 1e-12)).contiguous(),
                             1 + 0.1 * torch.randn(N, device=DEV, generator=g), 0.1 * torch.randn(N, device=DEV, generator=g))
    assert ops._lib.load().cldrd_gemm_nt_splitk_workspace(M, N, K) == 0        # the one-launch kernel takes these shapes: no partial sums
    outs = []
    for forced in (0, 1):
        ops.set_tuning("gemm_splitk", forced)
        out = torch.full((M + 3, N), 7.0, dtype=out_dtype, device=DEV)
        k2 = dict(kw)
        if k2.get("preact") is True:
            k2["preact"] = torch.full((M + 3, N), 7.0, dtype=torch.float16, device=DEV)
        try:
            ops.gemm_nt(A, B, out, M, **k2)
            torch.cuda.synchronize()
        finally:
            ops.set_tuning("gemm_splitk", 0)
        outs.append((out, k2.get("preact")))
    (o64, p64), (o128, p128) = outs
    assert torch.equal(o64, o128), (o64.float() - o128.float()).abs().max()
    assert (o64[M:] == 7.0).all() and not torch.isnan(o64.float()).any()
    if p64 is not None:
        assert torch.equal(p64, p128) and (p64[M:] == 7.0).all()
    if flavour == "plain":      # and against fp32 arithmetic
        ref = A[:M].float() @ B.float().T
        close(o64[:M].float(), ref, 2e-3, 2e-3 * ref.abs().max().item(), "64 x 64 kernel")


@pytest.mark.parametrize("M,N,K", [(256, 768, 3072), (240, 3072, 768), (130, 384, 1024), (512, 768, 768)])
@pytest.mark.parametrize("flavour", ["bias", "bias_gelu_dpre", "gelugrad_d", "res16", "bias_drop_res32_f32", "ln_res32_f32", "f16_bias_gelu", "f16_res32_f32"])
def test_gemm_nt_small_m_split_k(flavour, M, N, K):
    """Small-M problems are split along K (fp32 partials + a finishing launch with the same epilogue): against the one-pass kernel
    (ops.set_tuning("gemm_splitk", 1)) on the same inputs - 16-bit outputs agree to one rounding of the last place (the partial sums are added in a
    different order), fp32 outputs to 1e-5 of the row scale - and with an odd split count forced."""
    import oracle.dropout_ref as DRo
    f16 = flavour.startswith("f16")
    dt = torch.float16 if f16 else torch.bfloat16
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    A = (torch.randn(M, K, device=DEV, generator=g) * 0.5).to(dt)
    B = (torch.randn(N, K, device=DEV, generator=g) * 0.5).to(dt)
    bias = torch.randn(N, device=DEV, generator=g)
    kw, out_dtype = {}, dt
    if "bias" in flavour:
        kw["bias"] = bias
    if flavour == "bias_gelu_dpre":
        kw.update(preact=torch.empty(M, N, dtype=torch.bfloat16, device=DEV), act=3)
    if flavour == "f16_bias_gelu":
        kw.update(act=1)
    if flavour == "gelugrad_d":
        kw.update(gelu_pre=torch.rand(M, N, device=DEV, generator=g).bfloat16(), act=2)
    if flavour == "res16":
        kw["residual"] = torch.randn(M, N, device=DEV, generator=g).bfloat16()
    if "res32" in flavour:
        kw["residual"] = torch.randn(M, N, device=DEV, generator=g)
        out_dtype = torch.float32
        if flavour == "f16_res32_f32":
            kw["bias"] = bias
    if flavour == "bias_drop_res32_f32":
        kw.update(dropout_p=0.1, seed=(3 << 32) | 5)
    if flavour == "ln_res32_f32":
        s32 = kw["residual"]
        kw["bias"] = bias
        kw["residual_ln"] = (s32.mean(1).contiguous(), (1.0 / torch.sqrt(s32.var(1, unbiased=False) + 1e-12)).contiguous(),
                             1 + 0.1 * torch.randn(N, device=DEV, generator=g), 0.1 * torch.randn(N, device=DEV, generator=g))
    outs, pres = [], []
    for forced in (1, 0, 5):
        ops.set_tuning("gemm_splitk", forced)
        out = torch.full((M, N), float("nan"), dtype=out_dtype, device=DEV)
        if "preact" in kw:
            kw["preact"] = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        try:
            ops.gemm_nt(A, B, out, **kw)
            torch.cuda.synchronize()
        finally:
            ops.set_tuning("gemm_splitk", 0)
        outs.append(out.float())
        pres.append(kw["preact"].float() if "preact" in kw else None)
    ref = outs[0]
    assert not torch.isnan(ref).any()
    scale = ref.abs().max().item()
    for o, pr in zip(outs[1:], pres[1:]):
        if out_dtype == torch.float32:
            close(o, ref, 1e-5, 2e-5 * scale, f"split-K {flavour}")
        else:
            close(o, ref, 1 / 128, 1e-5 * scale, f"split-K {flavour}")          # one 16-bit rounding where the fp32 sums straddle a boundary
        if pr is not None:
            close(pr, pres[0], 1 / 128, 1e-4, "split-K saved derivative")
    if "drop" in flavour:          # the mask is keyed on (row, col) of the output: zeros in the same places
        assert torch.equal(outs[1] == kw["residual"], ref == kw["residual"])


def test_gemm_nt_dropout_is_deterministic_and_unbiased():
    M, N, K = 512, 256, 64
    A, B = bf(torch.ones(M, K)), bf(torch.ones(N, K) / K)
    out1 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    out2 = torch.empty_like(out1)
    ops.gemm_nt(A.to(DEV), B.to(DEV), out1, dropout_p=0.1, seed=1234)
    ops.gemm_nt(A.to(DEV), B.to(DEV), out2, dropout_p=0.1, seed=1234)
    assert torch.equal(out1, out2)
    keep = (out1 != 0).float().mean().item()
    assert abs(keep - 0.9) < 0.01
    kept = out1[out1 != 0]
    assert torch.allclose(kept, torch.full_like(kept, 1 / 0.9), rtol=1e-5)
    ops.gemm_nt(A.to(DEV), B.to(DEV), out2, dropout_p=0.1, seed=99)
    assert not torch.equal(out1, out2)


# The large-M ring kernel: every compiled epilogue flavour, BN = 192 and 256, more tiles than CUs, a partial last M tile (predicated
# epilogue), against torch fp32 on the same bf16 inputs.  (Until round 3 this also drove the persistent tile-walk kernel, removed in round 4.)
_PERS_FLAVOURS = ["plain", "bias", "bias_gelu_pre", "bias_gelu_dpre", "bias_gelu", "bias_res16", "bias_drop_res16", "gelugrad",
                  "gelugrad_d", "res16", "f32", "bias_res32_f32", "bias_drop_res32_f32"]


@pytest.mark.parametrize("M,N,K", [(16640, 768, 256), (20000, 1024, 192), (9000, 2304, 128)])
@pytest.mark.parametrize("flavour", _PERS_FLAVOURS)
def test_gemm_nt_ring_flavours_partial_tiles(flavour, M, N, K):
    import oracle.dropout_ref as DRo
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    A = (torch.randn(M, K, device=DEV, generator=g) * 0.5).bfloat16()
    B = (torch.randn(N, K, device=DEV, generator=g) * 0.5).bfloat16()
    bias = torch.randn(N, device=DEV, generator=g)
    res16 = torch.randn(M, N, device=DEV, generator=g).bfloat16()
    res32 = torch.randn(M, N, device=DEV, generator=g)
    gp = torch.randn(M, N, device=DEV, generator=g).bfloat16()
    base = A.float() @ B.float().T
    p_drop, seed = 0.1, (5 << 32) | 77
    kw, ref, out_dtype = {}, base, torch.bfloat16
    if flavour in ("f32", "bias_res32_f32", "bias_drop_res32_f32"):
        out_dtype = torch.float32
    if flavour.startswith("bias"):
        kw["bias"] = bias
        ref = ref + bias
    pre = None
    if flavour == "bias_gelu_pre":
        pre = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        kw.update(preact=pre, act=1)
    if flavour == "bias_gelu_dpre":
        pre = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        kw.update(preact=pre, act=3)
    if flavour == "bias_gelu":
        kw.update(act=1)
    if "gelu" in flavour and not flavour.startswith("gelugrad"):
        pre_ref = ref
        if flavour == "bias_gelu_dpre":
            pre_ref = 0.5 * (1 + torch.erf(ref / math.sqrt(2))) + ref * torch.exp(-0.5 * ref * ref) / math.sqrt(2 * math.pi)
        ref = torch.nn.functional.gelu(ref)
    if flavour == "gelugrad_d":
        ref = ref * gp.float()
        kw.update(gelu_pre=gp, act=2)
    if flavour == "gelugrad":
        x = gp.float()
        ref = ref * (0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi))
        kw["gelu_pre"] = gp
    if "drop" in flavour:
        keep = torch.from_numpy(DRo.keep_mask(seed, p_drop, M, N)).to(DEV)
        ref = torch.where(keep, ref / (1 - p_drop), torch.zeros((), device=DEV))
        kw.update(dropout_p=p_drop, seed=seed)
    if "res16" in flavour:
        kw["residual"] = res16
        ref = ref + res16.float()
    if "res32" in flavour:
        kw["residual"] = res32
        ref = ref + res32
    out = torch.full((M + 8, N), float("nan"), dtype=out_dtype, device=DEV)
    ops.gemm_nt(A, B, out, M, **kw)
    torch.cuda.synchronize()
    assert torch.isnan(out[M:].float()).all(), "rows past M were written"
    if out_dtype == torch.float32:
        close(out[:M], ref, 2e-5, 2e-4 * math.sqrt(K), flavour)
    else:
        close(out[:M], ref, 1 / 128, 2e-2, flavour)
    if pre is not None:
        close(pre, pre_ref, 1 / 128, 2e-2, "preact")
    # and the same bits from a second launch
    out2 = torch.full((M + 8, N), float("nan"), dtype=out_dtype, device=DEV)
    ops.gemm_nt(A, B, out2, M, **kw)
    torch.cuda.synchronize()
    assert torch.equal(out[:M], out2[:M]), f"{flavour}: two launches differ"


def test_gemm_nt_ring_repeated_launches_are_stable():
    """Race screen for the counted-vmcnt LDS-DMA ring: 30 launches of a 600-tile problem must agree bit for bit."""
    M, N, K = 256 * 200, 768, 768
    g = torch.Generator(device=DEV).manual_seed(3)
    A = torch.randn(M, K, device=DEV, generator=g).bfloat16()
    B = (torch.randn(N, K, device=DEV, generator=g) * 0.05).bfloat16()
    bias = torch.randn(N, device=DEV, generator=g)
    res = torch.randn(M, N, device=DEV, generator=g)
    first = torch.empty(M, N, device=DEV)
    ops.gemm_nt(A, B, first, bias=bias, residual=res)
    ref = A.float() @ B.float().T + bias + res
    close(first, ref, 2e-5, 2e-4 * math.sqrt(K), "200 row panels")
    for _ in range(30):
        out = torch.empty(M, N, device=DEV)
        ops.gemm_nt(A, B, out, bias=bias, residual=res)
        assert torch.equal(out, first)


# ------------------------------------------------------------------------------------------------ weight gradient
@pytest.mark.parametrize("M,N1,N2", [(64, 128, 128), (200, 128, 256), (1024, 384, 128), (4096, 768, 768), (3000, 256, 768),
                                     (8192, 2304, 768), (2048, 768, 3072), (130, 512, 128),
                                     # 256 x 192 tile (chosen from 24 such tiles up; forced below): token tails, one K tile, N2 = 192
                                     (130, 256, 192), (4000, 512, 384), (777, 256, 768), (64, 1536, 768), (5000, 1536, 768)])
def test_wgrad(M, N1, N2):
    Mp = ops.pad_rows(M)
    # rows >= M hold garbage (NaN): the kernel must not read them (its last K tile takes them from a zero page)
    dY, X = torch.full((Mp, N1), float("nan")), torch.full((Mp, N2), float("nan"))
    dY[:M], X[:M] = rnd(8, (M, N1)), rnd(9, (M, N2))
    dY, X = bf(dY), bf(X)
    ref = dY[:M].float().T @ X[:M].float()
    dW = torch.full((N1, N2), 7.0, dtype=torch.float32, device=DEV)
    ws = torch.empty(ops.wgrad_workspace_elems(M, N1, N2), dtype=torch.float32, device=DEV)
    ops.wgrad(dY.to(DEV), X.to(DEV), dW, M, ws, accumulate=False)
    close(dW, ref, 1e-4, 1e-4 * math.sqrt(M), f"wgrad {M}x{N1}x{N2}")
    db = torch.full((N1,), 3.0, dtype=torch.float32, device=DEV)
    ops.wgrad(dY.to(DEV), X.to(DEV), dW, M, ws, accumulate=True, dbias=db)
    close(dW, 2 * ref, 1e-4, 2e-4 * math.sqrt(M), "wgrad accumulate")
    close(db, 3.0 + dY[:M].double().sum(0), 1e-4, 1e-4 * math.sqrt(M), "wgrad fused bias gradient")


@pytest.mark.parametrize("accumulate", [False, True])
@pytest.mark.parametrize("shapes", [
    [(4100, 768, 768), (4096, 2304, 768), (4096, 3072, 768), (4096, 768, 3072), (30, 768, 768)],       # encoder layer + a short problem: 256 x 256 tiles (round 4), token tails
    [(3000, 768, 576), (3000, 1536, 768), (777, 512, 192)],                                            # N2 a multiple of 192 only: 256 x 192 tiles
    [(200, 128, 256), (200, 256, 128), (77, 128, 128)],                                                # tiny model: 128 x 128 tiles, token splits
    [(1000, 256, 192)] * 40,                                                                            # more than 32 problems: two launches
])
def test_wgrad_group(shapes, accumulate):
    """cldrd_wgrad_group: many problems, one launch; unpadded operands with token tails; bias gradients on some problems."""
    q = ops.WgradQueue()
    refs = []
    for i, (M, N1, N2) in enumerate(shapes):
        dY, X = bf(rnd(20 + i, (M, N1))), bf(rnd(60 + i, (M, N2)))
        dW = torch.full((N1, N2), 2.0, dtype=torch.float32, device=DEV)
        db = torch.full((N1,), -1.0, dtype=torch.float32, device=DEV) if i % 2 == 0 else None
        q.add(dY.to(DEV), X.to(DEV), dW, M, dbias=db)
        refs.append((dW, db, dY.float().T @ X.float(), dY.double().sum(0), M))
    q.flush(accumulate=accumulate)
    assert len(q) == 0
    for dW, db, rW, rb, M in refs:
        base = 2.0 if accumulate else 0.0
        close(dW, base + rW, 1e-4, 1e-4 * math.sqrt(M), "wgrad_group dW")
        if db is not None:
            close(db, (-1.0 if accumulate else 0.0) + rb, 1e-4, 1e-4 * math.sqrt(M), "wgrad_group dbias")


def test_wgrad_asymmetric():
    M, N1, N2 = 128, 128, 128
    dY = torch.zeros(M, N1)
    dY[torch.arange(M), torch.arange(M)] = 1.0            # identity -> dW = X (row i of dW = row i of X)
    X = (torch.arange(M * N2, dtype=torch.float32).reshape(M, N2) % 127) - 60.0
    dW = torch.empty(N1, N2, dtype=torch.float32, device=DEV)
    ws = torch.empty(ops.wgrad_workspace_elems(M, N1, N2), dtype=torch.float32, device=DEV)
    ops.wgrad(bf(dY).to(DEV), bf(X).to(DEV), dW, M, ws)
    assert torch.equal(dW.cpu(), bf(X).float())


# ------------------------------------------------------------------------------------------------ attention
def attn_ref(qkv, mask, nseq, L, H):
    d = H * 64
    x = qkv.double().view(nseq, L, 3, H, 64)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    s = q @ k.transpose(2, 3) * 0.125
    if mask is not None:
        s = s.masked_fill(mask[:, None, None, :] == 0, -1e30)
    p = torch.softmax(s, -1)
    lse = torch.logsumexp(s, -1)
    return (p @ v).transpose(1, 2).reshape(nseq * L, d), lse


@pytest.mark.parametrize("nseq,L,H,masked", [(3, 32, 2, False), (2, 30, 2, True), (4, 128, 2, True), (2, 100, 3, True),
                                              (2, 64, 12, False), (2, 256, 2, True), (1, 160, 1, True), (5, 8, 1, True),
                                              (1, 200, 2, True), (2, 129, 1, False), (1, 224, 1, True)])
def test_attention_fwd_bwd(nseq, L, H, masked):
    d = H * 64
    T = nseq * L
    qkv = bf(rnd(10, (T, 3 * d), 1.0))
    mask = None
    if masked:
        lens = np.clip(syn.msmarco_lengths(11, nseq, L), 2, L)
        lens[0] = L
        mask = torch.from_numpy((np.arange(L)[None, :] < lens[:, None]).astype(np.int64))
    qv = qkv.double().requires_grad_(True)
    ref, ref_lse = attn_ref(qv, mask, nseq, L, H)
    dctx = bf(rnd(12, (T, d)))
    ref.backward(dctx.double())
    ctx = torch.empty(T, d, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(nseq, H, L, dtype=torch.float32, device=DEV)
    md = None if mask is None else mask.to(DEV)
    ops.attention_fwd(qkv.to(DEV), md, ctx, lse, nseq, L, H)
    # P is rounded to bf16 before P.V (rel 2^-9 per term) and ctx is stored in bf16
    close(ctx, ref, 1 / 64, 2e-2, f"attention fwd L={L}")
    close(lse, ref_lse, 1e-4, 1e-3, "lse")
    dqkv = torch.zeros(T, 3 * d, dtype=torch.bfloat16, device=DEV)
    ops.attention_bwd(qkv.to(DEV), md, ctx, dctx.to(DEV), lse, dqkv, nseq, L, H)
    g = qv.grad.float()
    if mask is not None:      # gradients of padded query rows are not defined by the contract (they never reach the loss)
        valid = mask.bool().reshape(-1)
        # padded rows still receive dK/dV = 0 in the reference; the kernel writes them as 0 too
    scale = g.abs().max().item()
    close(dqkv, g, 1 / 32, 2e-2 * scale, f"attention bwd L={L}")


@pytest.mark.parametrize("nseq,L,H,fwd2", [(64, 128, 12, "1"), (64, 128, 12, "0"), (3, 40, 2, "1"), (2, 200, 2, "1"), (1, 256, 3, "1")])
def test_attention_fwd_writes_an_fp16_copy_of_its_context(nseq, L, H, fwd2, request):
    """`ctx16`: the context of a bf16 pass a second time in fp16 (the out-projection's operand), from the persistent kernel, the
    one-item-per-workgroup kernel (ops.set_tuning("attn_fwd2", 0)), the streaming kernel (L > 128) and the CLS-only kernel; with `ctx = None` only the
    fp16 tensor is written.  Both are roundings of the same fp32 value: the bf16 one is unchanged bit for bit, the fp16 one 8x closer."""
    ops.set_tuning("attn_fwd2", int(fwd2))
    request.addfinalizer(lambda: ops.set_tuning("attn_fwd2", 1))
    d, T = H * 64, nseq * L
    qkv = bf(rnd(150, (T, 3 * d), 1.0)).to(DEV)
    lens = np.clip(syn.msmarco_lengths(151, nseq, L), 2, L)
    lens[0] = L
    mask = torch.from_numpy((np.arange(L)[None, :] < lens[:, None]).astype(np.int64)).to(DEV)
    ref, _ = attn_ref(qkv.cpu().double(), mask.cpu(), nseq, L, H)
    plain = torch.empty(T, d, dtype=torch.bfloat16, device=DEV)
    ops.attention_fwd(qkv, mask, plain, None, nseq, L, H)
    ctx = torch.full((T, d), float("nan"), dtype=torch.bfloat16, device=DEV)
    c16 = torch.full((T, d), float("nan"), dtype=torch.float16, device=DEV)
    ops.attention_fwd(qkv, mask, ctx, None, nseq, L, H, ctx16=c16)
    assert torch.equal(ctx, plain)
    close(c16, ref, 1 / 64, 2e-2, "fp16 context copy")                       # P still goes through bf16
    assert (c16.float() - ctx.float()).abs().max().item() <= 2.0 ** -7 * ctx.float().abs().max().item()
    only = torch.full((T, d), float("nan"), dtype=torch.float16, device=DEV)
    ops.attention_fwd(qkv, mask, None, None, nseq, L, H, ctx16=only)
    assert torch.equal(only, c16)
    # CLS-only form
    kv = qkv[:, d:].contiguous()
    qc = qkv.view(nseq, L, 3 * d)[:, 0, :d].contiguous()
    probs = torch.empty(nseq, H, L, dtype=torch.float32, device=DEV)
    cb, ch = torch.empty(nseq, d, dtype=torch.bfloat16, device=DEV), torch.full((nseq, d), float("nan"), dtype=torch.float16, device=DEV)
    ops.attention_cls_fwd(qc, kv, mask, cb, probs, nseq, L, H, ctx16=ch)
    close(ch, ref.view(nseq, L, d)[:, 0], 1 / 256, 5e-3, "fp16 CLS context copy")
    ch2 = torch.full((nseq, d), float("nan"), dtype=torch.float16, device=DEV)
    ops.attention_cls_fwd(qc, kv, mask, None, probs, nseq, L, H, ctx16=ch2)
    assert torch.equal(ch, ch2)
    with pytest.raises(ValueError):
        ops.attention_fwd(qkv.to(torch.float16), mask, torch.empty(T, d, dtype=torch.float16, device=DEV), None, nseq, L, H, ctx16=c16)


@pytest.mark.parametrize("nseq,L,H,p", [(48, 128, 12, 0.0), (48, 128, 12, 0.1), (90, 100, 6, 0.1), (200, 30, 3, 0.0), (70, 64, 8, 0.1),
                                         (43, 96, 12, 0.0)])
def test_attention_bwd_persistent_two_role_kernel(nseq, L, H, p):
    """>= 2 items per CU and L <= 128: the persistent kernel (sweep A and sweep B on different waves, next item prefetched through
    registers into the second LDS buffer).  Same arithmetic in the same order as the one-item-per-workgroup kernel, which the
    reference tests above pin: the two must agree bit for bit, masked and ragged, with and without dropout, over several items per
    workgroup (a stale or half-written LDS buffer would show up here)."""
    d, T = H * 64, nseq * L
    g = torch.Generator(device=DEV).manual_seed(nseq * L + H)
    qkv = torch.randn(T, 3 * d, device=DEV, generator=g).bfloat16()
    dctx = torch.randn(T, d, device=DEV, generator=g).bfloat16()
    lens = torch.randint(2, L + 1, (nseq,), device=DEV, generator=g)
    lens[0] = L
    mask = (torch.arange(L, device=DEV)[None, :] < lens[:, None]).to(torch.int64).contiguous()
    ctx = torch.empty(T, d, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(nseq, H, L, dtype=torch.float32, device=DEV)
    bits = ops.attention_drop_bits(nseq, L, H, p, DEV)          # the forward's dropout keep bits (None without dropout)
    assert (bits is not None) == (p > 0)
    if bits is not None:
        bits.fill_(-1)
    ops.attention_fwd(qkv, mask, ctx, lse, nseq, L, H, dropout_p=p, seed=99, drop_bits=bits)
    outs = []
    for env, b in (("0", None), (None, None), (None, None), (None, bits)):
        ops.set_tuning("attn_bwd2", 0 if env == "0" else 1)
        dqkv = torch.full((T, 3 * d), float("nan"), dtype=torch.bfloat16, device=DEV)
        try:
            ops.attention_bwd(qkv, mask, ctx, dctx, lse, dqkv, nseq, L, H, dropout_p=p, seed=99, drop_bits=b)
            torch.cuda.synchronize()
        finally:
            ops.set_tuning("attn_bwd2", 1)
        outs.append(dqkv)
    valid = mask.bool().reshape(-1)
    assert not torch.isnan(outs[1][valid].float()).any()
    assert torch.equal(outs[0][valid], outs[1][valid]), "persistent kernel differs from the one-item-per-workgroup kernel"
    assert torch.equal(outs[1], outs[2]), "two launches of the persistent kernel differ"
    assert torch.equal(outs[1], outs[3]), "the forward's keep bits give a different mask than the hash"


@pytest.mark.parametrize("nseq,L,H,p", [(48, 128, 12, 0.0), (48, 128, 12, 0.1), (90, 100, 6, 0.1), (200, 30, 3, 0.0), (70, 64, 8, 0.1)])
def test_attention_fwd_persistent_loader_kernel(nseq, L, H, p):
    """>= 2 items per CU and L <= 128: the persistent forward (4 compute waves + 4 loader waves, double-buffered LDS) must reproduce the
    one-item-per-workgroup kernel bit for bit (context and LSE), which the reference tests above pin."""
    d, T = H * 64, nseq * L
    g = torch.Generator(device=DEV).manual_seed(nseq * L + H + 1)
    qkv = torch.randn(T, 3 * d, device=DEV, generator=g).bfloat16()
    lens = torch.randint(2, L + 1, (nseq,), device=DEV, generator=g)
    lens[0] = L
    mask = (torch.arange(L, device=DEV)[None, :] < lens[:, None]).to(torch.int64).contiguous()
    outs = []
    for env in ("0", None, "bits"):
        ops.set_tuning("attn_fwd2", 0 if env == "0" else 1)
        ctx = torch.full((T, d), float("nan"), dtype=torch.bfloat16, device=DEV)
        lse = torch.full((nseq, H, L), float("nan"), dtype=torch.float32, device=DEV)
        try:
            bits = ops.attention_drop_bits(nseq, L, H, p, DEV) if env == "bits" else None
            ops.attention_fwd(qkv, mask, ctx, lse, nseq, L, H, dropout_p=p, seed=1234, drop_bits=bits)
            torch.cuda.synchronize()
        finally:
            ops.set_tuning("attn_fwd2", 1)
        outs.append((ctx, lse))
    assert not torch.isnan(outs[1][0].float()).any() and not torch.isnan(outs[1][1]).any()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), "persistent forward differs from the one-item kernel"
    assert torch.equal(outs[1][0], outs[2][0]) and torch.equal(outs[1][1], outs[2][1])


@pytest.mark.parametrize("nseq,L,H,p,f16,masked", [(48, 256, 12, 0.0, False, True), (48, 256, 12, 0.1, True, True), (90, 200, 6, 0.1, False, True),
                                                     (60, 130, 12, 0.0, True, False), (64, 161, 8, 0.0, False, True), (260, 256, 2, 0.1, True, True),
                                                     (70, 250, 8, 0.0, False, False)])
def test_attention_fwd_streaming_persistent_kernel(nseq, L, H, p, f16, masked):
    """128 < L <= 256 with >= 2 items per CU (cfg5 index encode at max_length 256, cfg4 training): the persistent streaming forward (K / V
    double-buffered in LDS, Q straight from global memory, next item prefetched through registers) must reproduce the one-item-per-workgroup
    streaming kernel bit for bit - context, fp16 context copy and LSE; masked, unmasked, ragged, odd key-block counts, bf16 and fp16 families,
    with and without dropout, several items per workgroup (a stale or half-written LDS buffer, or a Q fragment of the wrong item, shows here)."""
    d, T = H * 64, nseq * L
    g = torch.Generator(device=DEV).manual_seed(nseq * L + H + 7)
    dt = torch.float16 if f16 else torch.bfloat16
    qkv = torch.randn(T, 3 * d, device=DEV, generator=g).to(dt)
    mask = None
    if masked:
        lens = torch.randint(2, L + 1, (nseq,), device=DEV, generator=g)
        lens[0] = L
        mask = (torch.arange(L, device=DEV)[None, :] < lens[:, None]).to(torch.int64).contiguous()
    outs = []
    for tune in (0, 1, 1):
        ops.set_tuning("attn_fwd2", tune)
        ctx = torch.full((T, d), float("nan"), dtype=dt, device=DEV)
        c16 = None if f16 else torch.full((T, d), float("nan"), dtype=torch.float16, device=DEV)
        lse = torch.full((nseq, H, L), float("nan"), dtype=torch.float32, device=DEV)
        try:
            ops.attention_fwd(qkv, mask, ctx, lse, nseq, L, H, dropout_p=p, seed=4321, ctx16=c16, full_family=f16)
            torch.cuda.synchronize()
        finally:
            ops.set_tuning("attn_fwd2", 1)
        outs.append((ctx, lse, c16))
    valid = mask.bool().reshape(-1) if masked else torch.ones(T, dtype=torch.bool, device=DEV)
    assert not torch.isnan(outs[1][0][valid].float()).any()
    lv = mask.bool()[:, None, :].expand(nseq, H, L) if masked else torch.ones(nseq, H, L, dtype=torch.bool, device=DEV)
    assert torch.equal(outs[0][0][valid], outs[1][0][valid]), "persistent streaming forward differs from the one-item kernel (context)"
    assert torch.equal(outs[0][1][lv], outs[1][1][lv]), "persistent streaming forward differs from the one-item kernel (LSE)"
    if c16 is not None:
        assert torch.equal(outs[0][2][valid], outs[1][2][valid])
    assert torch.equal(outs[1][0][valid], outs[2][0][valid]) and torch.equal(outs[1][1][lv], outs[2][1][lv]), "two launches differ"
    # and against the fp64 reference (the one-item kernel is pinned elsewhere; this is the new kernel's own check)
    if p == 0.0:
        ref, ref_lse = attn_ref(qkv.cpu().double(), mask.cpu() if masked else None, nseq, L, H)
        close(outs[1][0][valid], ref.to(DEV)[valid], 1 / 64, 2e-2, "streaming persistent forward vs fp64")


@pytest.mark.parametrize("nseq,L,H,p,f16", [(48, 128, 12, 0.1, True), (48, 128, 12, 0.0, False), (90, 100, 6, 0.1, False), (200, 30, 3, 0.1, True),
                                             (7, 128, 12, 0.1, True), (5, 77, 2, 0.0, False), (48, 256, 12, 0.1, True), (64, 161, 8, 0.0, False),
                                             (6, 256, 12, 0.1, False), (9, 200, 4, 0.1, True)])
def test_attention_on_packed_rows_equals_the_padded_layout(nseq, L, H, p, f16):
    """Round 6: cldrd_attention_{fwd,bwd,cls_fwd,cls_bwd}_varlen read and write the PACKED rows of a batch through cu (sequence m = rows cu[m] ..
    cu[m + 1]) instead of the padded [nseq * L, .] layout the encoder used to move them to and from (cldrd_unpack_rows16 / cldrd_gather_rows: 0.56 ms of
    an 8.3-ms MS MARCO-shaped training step).  Same kernels, same arithmetic, zero rows where the padded layout held zeros: context, LSE, keep bits,
    q / k / v gradients, CLS context, probabilities and K / V gradients must equal the padded path bit for bit - few and many items (the
    one-item-per-workgroup and the persistent kernels), L <= 128 and 128 < L <= 256, bf16 and fp16 families, with and without dropout; lengths include
    1 and L; a packed buffer exactly Tp rows long (a read or write past a sequence's rows lands in the next sequence or past the end)."""
    d = H * 64
    g = torch.Generator(device=DEV).manual_seed(nseq * L + H + 13)
    dt = torch.float16 if f16 else torch.bfloat16
    lens = torch.randint(1, L + 1, (nseq,), device=DEV, generator=g)
    lens[0], lens[-1] = L, 1
    cu = torch.zeros(nseq + 1, dtype=torch.int32, device=DEV)
    cu[1:] = torch.cumsum(lens, 0).to(torch.int32)
    Tp = int(cu[-1])
    mask = (torch.arange(L, device=DEV)[None, :] < lens[:, None]).to(torch.int64).contiguous()
    tok = torch.nonzero(mask.reshape(-1)).reshape(-1).to(torch.int32)          # padded row of every packed row
    qkv_p = torch.randn(Tp, 3 * d, device=DEV, generator=g).to(dt)
    dctx_p = torch.randn(Tp, d, device=DEV, generator=g).to(dt)

    def padded(t_p):
        out = torch.full((nseq * L, t_p.shape[1]), float("nan"), dtype=dt, device=DEV)
        ops.unpack_rows16(t_p, out, cu, nseq, L)
        return out

    def packed(t_pad):
        out = torch.empty(Tp, t_pad.shape[1], dtype=t_pad.dtype, device=DEV)
        ops.gather_rows(t_pad, tok, out, Tp)
        return out
    qkv = padded(qkv_p)
    res = {}
    for kind in ("padded", "packed"):
        pk = kind == "packed"
        rows = Tp if pk else nseq * L
        ctx = torch.full((rows, d), float("nan"), dtype=dt, device=DEV)
        c16 = None if f16 else torch.full((rows, d), float("nan"), dtype=torch.float16, device=DEV)
        lse = torch.full((nseq, H, L), float("nan"), dtype=torch.float32, device=DEV)
        bits = ops.attention_drop_bits(nseq, L, H, p, DEV)
        ops.attention_fwd(qkv_p if pk else qkv, None if pk else mask, ctx, lse, nseq, L, H, dropout_p=p, seed=77, drop_bits=bits, ctx16=c16,
                          full_family=f16, cu=cu if pk else None)
        dqkv = torch.full((rows, 3 * d), float("nan"), dtype=dt, device=DEV)
        ops.attention_bwd(qkv_p if pk else qkv, None if pk else mask, ctx if pk else ctx, dctx_p if pk else padded(dctx_p), lse, dqkv, nseq, L, H,
                          dropout_p=p, seed=77, drop_bits=bits, cu=cu if pk else None)
        # the CLS-only last layer: queries of token 0, K | V of every token
        qc = (qkv_p[cu[:-1].long(), :d] if pk else qkv.view(nseq, L, 3 * d)[:, 0, :d]).contiguous()
        kv = (qkv_p if pk else qkv)[:, d:].contiguous()
        cctx = torch.full((nseq, d), float("nan"), dtype=dt, device=DEV)
        probs = torch.full((nseq, H, L), float("nan"), dtype=torch.float32, device=DEV)
        ops.attention_cls_fwd(qc, kv, None if pk else mask, cctx, probs, nseq, L, H, dropout_p=p, seed=78, cu=cu if pk else None)
        dqc = torch.full((nseq, d), float("nan"), dtype=dt, device=DEV)
        dkv = torch.full((rows, 2 * d), float("nan"), dtype=dt, device=DEV)
        ops.attention_cls_bwd(qc, kv, probs, dctx_p[cu[:-1].long()].contiguous(), dqc, dkv, nseq, L, H, dropout_p=p, seed=78, cu=cu if pk else None)
        torch.cuda.synchronize()
        if not pk:
            ctx, dqkv, dkv = packed(ctx), packed(dqkv), packed(dkv)
            c16 = packed(c16) if c16 is not None else None
        res[kind] = dict(ctx=ctx, c16=c16, lse=lse, bits=bits, dqkv=dqkv, cctx=cctx, probs=probs, dqc=dqc, dkv=dkv)
    a, b = res["padded"], res["packed"]
    lv = mask.bool()[:, None, :].expand(nseq, H, L)
    for k in ("ctx", "c16", "dqkv", "cctx", "dqc", "dkv"):
        if a[k] is None:
            continue
        assert not torch.isnan(b[k].float()).any(), k
        assert torch.equal(a[k], b[k]), f"{k}: the packed rows give other bits than the padded layout"
    assert torch.equal(a["lse"][lv], b["lse"][lv]) and torch.equal(a["probs"][lv], b["probs"][lv])
    assert (b["probs"][~lv] == 0).all()
    if a["bits"] is not None:
        assert torch.equal(a["bits"], b["bits"])


@pytest.mark.parametrize("L,H,f16,p", [(128, 1, False, 0.0), (128, 4, False, 0.0), (128, 4, True, 0.1), (128, 1, True, 0.1), (100, 2, False, 0.0), (64, 1, False, 0.0),
                                       (256, 1, False, 0.0), (256, 2, True, 0.0), (256, 2, False, 0.1), (200, 1, True, 0.0)])
def test_attention_on_packed_rows_every_length(L, H, f16, p):
    """One launch holds a sequence of EVERY length 1 .. L (in a shuffled order), so every count of live / skipped 32-key blocks and every position of
    the last real key inside a block occurs, in the one-item-per-workgroup kernels (H = 1 or 2: L or 2 L items) and in the persistent ones (4 L items
    >= two per CU): packed rows against the padded layout, bit for bit, forward and backward.  This is the test the block skipping needed: hipcc left
    an MFMA -> accumulator-read hazard open on the early-exit edge of a block loop, and only sequences of 59 .. 64 tokens at L = 128 showed it (the
    stale registers held keys 58, 59, 62, 63, masked in anything shorter; tools/wgrad_attn_fuzz.py found it, mfma_drain() in attention.hip closes it)."""
    nseq, d = L, H * 64
    g = torch.Generator(device=DEV).manual_seed(L * 7 + H)
    dt = torch.float16 if f16 else torch.bfloat16
    lens = (torch.randperm(L, device=DEV, generator=g) + 1).to(torch.int64)
    cu = torch.zeros(nseq + 1, dtype=torch.int32, device=DEV)
    cu[1:] = torch.cumsum(lens, 0).to(torch.int32)
    Tp = int(cu[-1])
    mask = (torch.arange(L, device=DEV)[None, :] < lens[:, None]).to(torch.int64).contiguous()
    tok = torch.nonzero(mask.reshape(-1)).reshape(-1)
    qkv = torch.randn(nseq * L, 3 * d, device=DEV, generator=g).to(dt)            # padded layout with REAL values in the padding rows (masked, not zero)
    dctx = torch.randn(nseq * L, d, device=DEV, generator=g).to(dt) * mask.reshape(-1, 1).to(dt)
    out = {}
    for pk in (False, True):
        rows = Tp if pk else nseq * L
        ctx = torch.full((rows, d), float("nan"), dtype=dt, device=DEV)
        lse = torch.full((nseq, H, L), float("nan"), dtype=torch.float32, device=DEV)
        bits = ops.attention_drop_bits(nseq, L, H, p, DEV)
        x = qkv[tok].contiguous() if pk else qkv
        ops.attention_fwd(x, None if pk else mask, ctx, lse, nseq, L, H, dropout_p=p, seed=31, drop_bits=bits, full_family=f16, cu=cu if pk else None)
        dqkv = torch.full((rows, 3 * d), float("nan"), dtype=dt, device=DEV)
        ops.attention_bwd(x, None if pk else mask, ctx, dctx[tok].contiguous() if pk else dctx, lse, dqkv, nseq, L, H, dropout_p=p, seed=31, drop_bits=bits,
                          cu=cu if pk else None)
        torch.cuda.synchronize()
        out[pk] = (ctx if pk else ctx[tok], dqkv if pk else dqkv[tok], lse)
    lv = mask.bool()[:, None, :].expand(nseq, H, L)
    seq_of = torch.repeat_interleave(torch.arange(nseq, device=DEV), lens)
    for i, name in ((0, "context"), (1, "q / k / v gradients")):
        same = (out[True][i] == out[False][i]).all(1)
        bad = sorted(set(lens[seq_of[~same]].tolist()))
        assert not bad, f"{name}: packed rows differ from the padded layout for sequences of {bad[:20]} tokens"
    assert torch.equal(out[True][2][lv], out[False][2][lv]) and bool(torch.isfinite(out[True][2]).all())


@pytest.mark.parametrize("L,H,f16,p", [(256, 2, False, 0.0), (256, 4, True, 0.0), (200, 1, True, 0.0)])
def test_attention_sequence_lists_send_short_sequences_through_the_short_kernels(L, H, f16, p):
    """cldrd_attention_{fwd,bwd}_varlen_list: a packed batch at 128 < L <= 256 in TWO launches - the sequences of at most 128 tokens through the
    L <= 128 kernels (tile height 128), the longer ones through the L <= 256 kernels - with LSE rows and dropout row keys on the stride L of the
    whole batch.  One batch holds a sequence of every length 1 .. L.  Without dropout: the long sequences equal the one-launch result bit for bit
    (same kernels), the short ones equal a launch of the SAME sequences as a batch of their own at L = 128 bit for bit (same kernels, other
    strides), context, LSE and q / k / v gradients.  (With dropout: test_attention_sequence_lists_with_dropout_against_the_oracle_mask.)"""
    nseq, d = L, H * 64
    g = torch.Generator(device=DEV).manual_seed(L * 3 + H)
    dt = torch.float16 if f16 else torch.bfloat16
    lens = (torch.randperm(L, device=DEV, generator=g) + 1).to(torch.int64)
    cu = torch.zeros(nseq + 1, dtype=torch.int32, device=DEV)
    cu[1:] = torch.cumsum(lens, 0).to(torch.int32)
    Tp = int(cu[-1])
    qkv = torch.randn(Tp, 3 * d, device=DEV, generator=g).to(dt)
    dctx = torch.randn(Tp, d, device=DEV, generator=g).to(dt)
    short = torch.nonzero(lens <= 128).reshape(-1).to(torch.int32)
    long_ = torch.nonzero(lens > 128).reshape(-1).to(torch.int32)
    assert short.numel() == 128 and long_.numel() == L - 128

    def run(groups, Lr=L, x=qkv, dy=dctx, cur=cu, n=nseq):
        rows = x.shape[0]
        ctx = torch.full((rows, d), float("nan"), dtype=dt, device=DEV)
        lse = torch.full((n, H, Lr), float("nan"), dtype=torch.float32, device=DEV)
        dq = torch.full((rows, 3 * d), float("nan"), dtype=dt, device=DEV)
        for sl, tile in groups:
            ops.attention_fwd(x, None, ctx, lse, n, Lr, H, dropout_p=p, seed=19, full_family=f16, cu=cur, seq_list=sl, tile=tile)
        for sl, tile in groups:
            ops.attention_bwd(x, None, ctx, dy, lse, dq, n, Lr, H, dropout_p=p, seed=19, cu=cur, seq_list=sl, tile=tile)
        torch.cuda.synchronize()
        return ctx, lse, dq
    one = run([(None, 0)])
    two = run([(short, 128), (long_, L)])
    assert not torch.isnan(two[0].float()).any() and not torch.isnan(two[2].float()).any()
    seq_of = torch.repeat_interleave(torch.arange(nseq, device=DEV), lens)
    is_long = (lens > 128)[seq_of]
    valid = torch.arange(L, device=DEV)[None, :] < lens[:, None]
    lv = valid[:, None, :].expand(nseq, H, L)
    if p == 0.0:
        assert torch.equal(two[0][is_long], one[0][is_long]) and torch.equal(two[2][is_long], one[2][is_long])
        longv = lv & (lens > 128)[:, None, None]
        assert torch.equal(two[1][longv], one[1][longv])
        # the short sequences as a batch of their own at L = 128
        sl = short.long()
        rows_s = torch.cat([torch.arange(int(cu[m]), int(cu[m + 1]), device=DEV) for m in sl.tolist()])
        cu_s = torch.zeros(sl.numel() + 1, dtype=torch.int32, device=DEV)
        cu_s[1:] = torch.cumsum(lens[sl], 0).to(torch.int32)
        own = run([(None, 0)], Lr=128, x=qkv[rows_s].contiguous(), dy=dctx[rows_s].contiguous(), cur=cu_s, n=sl.numel())
        assert torch.equal(two[0][rows_s], own[0]), "short sequences: context differs from the L = 128 launch of the same sequences"
        assert torch.equal(two[2][rows_s], own[2]), "short sequences: q / k / v gradients differ from the L = 128 launch"
        vs = (torch.arange(128, device=DEV)[None, :] < lens[sl][:, None])[:, None, :].expand(sl.numel(), H, 128)
        assert torch.equal(two[1][sl][:, :, :128][vs], own[1][vs])
    # everything within 16-bit rounding of the one-launch result (the short sequences change kernels: another summation order)
    tol = 2.0 ** (-9 if f16 else -6)
    for a, b, name in ((two[0], one[0], "context"), (two[2], one[2], "gradients")):
        err = (a.float() - b.float()).abs()
        assert bool((err <= tol * b.float().abs() + tol * b.float().abs().max() * 0.25).all()), (name, float(err.max()))
    assert torch.allclose(two[1][lv], one[1][lv], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("f16", [False, True])
def test_attention_sequence_lists_with_dropout_against_the_oracle_mask(f16):
    """The two-launch form of a packed L = 256 batch WITH dropout against fp64 and the mask of oracle/dropout_ref.py (a function of sequence,
    head, L, query, key: the short kernels must hash with the batch's L = 256, not with their tile height), forward and backward, same bars as
    test_attention_dropout_fwd_bwd_against_the_oracle_mask; and the one-launch form against the same reference."""
    p, seed, L, H = 0.25, 977, 256, 2
    lens_l = [256, 1, 17, 64, 100, 128, 129, 200, 33, 96, 127, 130, 255, 2, 59, 64, 65, 31, 32, 160, 8, 77, 78, 192]
    nseq, d = len(lens_l), H * 64
    dt = torch.float16 if f16 else torch.bfloat16
    lens = torch.tensor(lens_l, dtype=torch.int64, device=DEV)
    cu = torch.zeros(nseq + 1, dtype=torch.int32, device=DEV)
    cu[1:] = torch.cumsum(lens, 0).to(torch.int32)
    mask = (torch.arange(L, device=DEV)[None, :] < lens[:, None])
    tok = torch.nonzero(mask.reshape(-1)).reshape(-1)
    g = torch.Generator(device=DEV).manual_seed(5)
    qkv = torch.randn(nseq * L, 3 * d, device=DEV, generator=g).to(dt)
    dctx = (torch.randn(nseq * L, d, device=DEV, generator=g).to(dt) * mask.reshape(-1, 1).to(dt))
    keep = torch.from_numpy(DR.attention_keep_mask(seed, p, nseq, H, L)).to(DEV)
    qv = qkv.double().requires_grad_(True)
    x = qv.view(nseq, L, 3, H, 64)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    sc = (q @ k.transpose(2, 3) * 0.125).masked_fill(~mask[:, None, None, :], -1e30)
    ref = ((torch.softmax(sc, -1) * keep / (1 - p)) @ v).transpose(1, 2).reshape(nseq * L, d)
    ref.backward(dctx.double())
    gref = qv.grad[tok].float()
    short = torch.nonzero(lens <= 128).reshape(-1).to(torch.int32)
    long_ = torch.nonzero(lens > 128).reshape(-1).to(torch.int32)
    x_p, dy_p, Tp = qkv[tok].contiguous(), dctx[tok].contiguous(), int(cu[-1])
    for groups in ([(short, 128), (long_, L)], [(None, 0)]):
        ctx = torch.full((Tp, d), float("nan"), dtype=dt, device=DEV)
        lse = torch.full((nseq, H, L), float("nan"), dtype=torch.float32, device=DEV)
        dq = torch.full((Tp, 3 * d), float("nan"), dtype=dt, device=DEV)
        for sl, tile in groups:
            ops.attention_fwd(x_p, None, ctx, lse, nseq, L, H, dropout_p=p, seed=seed, full_family=f16, cu=cu, seq_list=sl, tile=tile)
        for sl, tile in groups:
            ops.attention_bwd(x_p, None, ctx, dy_p, lse, dq, nseq, L, H, dropout_p=p, seed=seed, cu=cu, seq_list=sl, tile=tile)
        close(ctx, ref.detach()[tok].float(), 1 / 64, 3e-2, f"packed attention fwd with dropout, {len(groups)} launch(es)")
        close(dq, gref, 1 / 32, 2e-2 * gref.abs().max().item(), f"packed attention bwd with dropout, {len(groups)} launch(es)")


def test_gather_i64_picks_the_token_ids_of_the_packed_rows():
    """cldrd_gather_i64: out[p] = src[idx[p]] for int64 elements (what torch.index_select + an index cast did for a packed batch's token ids)."""
    g = torch.Generator(device=DEV).manual_seed(12)
    src = torch.randint(-2 ** 40, 2 ** 40, (37, 129), device=DEV, generator=g, dtype=torch.int64)
    for n in (1, 63, 4773, 100000):
        idx = torch.randint(0, src.numel(), (n,), device=DEV, generator=g).to(torch.int32)
        assert torch.equal(ops.gather_i64(src.reshape(-1), idx), src.reshape(-1)[idx.long()])
    assert torch.equal(ops.gather_i64(src.reshape(-1), idx, 10), src.reshape(-1)[idx[:10].long()])
    with pytest.raises((TypeError, ValueError)):
        ops.gather_i64(src.reshape(-1).to(torch.int32), idx)


def test_attention_dropout_statistics():
    nseq, L, H = 2, 64, 2
    T, d = nseq * L, H * 64
    qkv = torch.zeros(T, 3 * d)
    qkv[:, 2 * d:] = 1.0                                  # V = 1 -> ctx = sum_k P_drop = kept mass / (1 - p)
    ctx = torch.empty(T, d, dtype=torch.bfloat16, device=DEV)
    ops.attention_fwd(bf(qkv).to(DEV), None, ctx, None, nseq, L, H, dropout_p=0.5, seed=7)
    m = ctx.float().mean().item()
    assert abs(m - 1.0) < 0.05
    ctx2 = torch.empty_like(ctx)
    ops.attention_fwd(bf(qkv).to(DEV), None, ctx2, None, nseq, L, H, dropout_p=0.5, seed=7)
    assert torch.equal(ctx, ctx2)


# ------------------------------------------------------------------------------------------------ LayerNorm / embeddings
@pytest.mark.parametrize("T,d", [(7, 128), (1000, 768), (130, 1024), (64, 256)])
def test_layernorm_fwd_bwd(T, d):
    x = bf(rnd(13, (T, d), 2.0) + 0.5)
    gamma, beta = 1 + rnd(14, (d,), 0.1), rnd(15, (d,), 0.1)
    dy = bf(rnd(16, (T, d)))
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    y = torch.nn.functional.layer_norm(xr, (d,), gr, br, 1e-12)
    y.backward(dy.double())
    out = torch.empty(T, d, dtype=torch.bfloat16, device=DEV)
    mean, rstd = torch.empty(T, device=DEV), torch.empty(T, device=DEV)
    L = 4 if T % 4 == 0 else 1
    cls = torch.empty((T + L - 1) // L, d, dtype=torch.float32, device=DEV)
    ops.layernorm_fwd(x.to(DEV), gamma.to(DEV), beta.to(DEV), out, mean, rstd, T, 1e-12, cls, L)
    close(out, y, 1 / 128, 1e-2, "ln fwd")
    close(cls[: T // L if T % L == 0 else T], y[::L], 1e-5, 1e-5, "ln cls fp32")
    dx = torch.empty(T, d, dtype=torch.bfloat16, device=DEV)
    dg, db, dbias = (torch.zeros(d, device=DEV) for _ in range(3))
    partial = torch.empty(ops.ln_partial_elems(T, d), device=DEV)
    ops.layernorm_bwd(dy.to(DEV), x.to(DEV), mean, rstd, gamma.to(DEV), dx, None, dg, db, dbias, partial, T, accumulate=False)
    close(dx, xr.grad, 1 / 64, 1e-2 * xr.grad.abs().max().item(), "ln dx")
    close(dg, gr.grad, 1e-3, 1e-3 * gr.grad.abs().max().item(), "ln dgamma")
    close(db, br.grad, 1e-3, 1e-3 * br.grad.abs().max().item(), "ln dbeta")
    close(dbias, dx.float().sum(0), 2e-2, 2e-2 * dx.float().sum(0).abs().max().item() + 1e-2, "ln dbias")


@pytest.mark.parametrize("T,d,p", [(1000, 768, 0.0), (130, 1024, 0.1), (64, 256, 0.1), (4097, 768, 0.1)])
def test_layernorm_bwd_fp32_gradient_stream(T, d, p):
    """fp32 dy in, fp32 dx out (the gradient stream is rounded nowhere), bf16 dx_dropped next to it (the MFMA operand of the next
    data-gradient GEMM): dx against the fp64 formula at fp32 accuracy, dx_dropped = bf16(masked dx) with the bf16-stream kernel's mask,
    parameter gradients as before."""
    x = rnd(113, (T, d), 2.0) + 0.5
    gamma = 1 + rnd(114, (d,), 0.1)
    dy = rnd(116, (T, d)) * 1e-3
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), torch.zeros(d, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.layer_norm(xr, (d,), gr, br, 1e-12).backward(dy.double())
    mean = x.mean(1).to(DEV)
    rstd = (1.0 / torch.sqrt(x.var(1, unbiased=False) + 1e-12)).to(DEV)
    dx = torch.full((T, d), float("nan"), dtype=torch.float32, device=DEV)
    dxm = torch.full((T, d), float("nan"), dtype=torch.bfloat16, device=DEV)
    dg, db, dbias = (torch.zeros(d, device=DEV) for _ in range(3))
    partial = torch.empty(ops.ln_partial_elems(T, d), device=DEV)
    ops.layernorm_bwd(dy.to(DEV), x.to(DEV), mean, rstd, gamma.to(DEV), dx, dxm, dg, db, dbias, partial, T, dropout_p=p, seed=77, accumulate=False)
    scale = xr.grad.abs().max().item()
    close(dx, xr.grad, 1e-4, 1e-5 * scale, "ln dx fp32")
    close(dg, gr.grad, 1e-3, 1e-3 * gr.grad.abs().max().item(), "ln dgamma")
    close(db, br.grad, 1e-3, 1e-3 * br.grad.abs().max().item(), "ln dbeta")
    # the same call on the bf16 stream gives the mask: where its dropped copy is zero and its dx is not
    dx16, dxm16 = torch.empty(T, d, dtype=torch.bfloat16, device=DEV), torch.empty(T, d, dtype=torch.bfloat16, device=DEV)
    ops.layernorm_bwd(bf(dy).to(DEV), x.to(DEV), mean, rstd, gamma.to(DEV), dx16, dxm16, None, None, None, partial, T, dropout_p=p, seed=77)
    keep = ~((dxm16.float() == 0) & (dx16.float() != 0))
    want = torch.where(keep, dx / (1.0 - p), torch.zeros_like(dx))
    close(dxm, want, 1 / 128, 1e-6 * scale, "dx_dropped = bf16(mask * dx / (1 - p))")
    assert torch.equal(dxm.float() == 0, want == 0), "dropout mask differs from the bf16-stream kernel's"
    close(dbias, want.sum(0), 1e-3, 1e-4 * scale * T ** 0.5, "ln dbias")
    with pytest.raises(ValueError):
        ops.layernorm_bwd(dy.to(DEV), x.to(DEV), mean, rstd, gamma.to(DEV), dx, None, dg, db, dbias, partial, T)       # operand copy required
    # dy = fp32 stream + bf16 branch, added on load: the same results as the sum handed over in one tensor
    branch = bf(rnd(117, (T, d)) * 1e-3)
    dxb = torch.empty_like(dx); dxmb = torch.empty_like(dxm)
    g3 = [torch.zeros(d, device=DEV) for _ in range(3)]
    ops.layernorm_bwd((dy - branch).to(DEV), x.to(DEV), mean, rstd, gamma.to(DEV), dxb, dxmb, g3[0], g3[1], g3[2], partial, T, dropout_p=p, seed=77,
                      accumulate=False, dy_branch=branch.to(DEV))
    close(dxb, dx, 1e-4, 2e-6 * scale, "ln dx (stream + branch)")
    close(g3[0], dg, 1e-4, 1e-5 * dg.abs().max().item(), "ln dgamma (stream + branch)")
    close(g3[1], db, 1e-4, 1e-5 * db.abs().max().item() + 1e-9, "ln dbeta (stream + branch)")
    with pytest.raises(ValueError):
        ops.layernorm_bwd(bf(dy).to(DEV), x.to(DEV), mean, rstd, gamma.to(DEV), dx16, dxm16, None, None, None, partial, T, dy_branch=branch.to(DEV))


def test_stream_row_ops_on_fp32_rows():
    """scatter_cls_grad(_idx) / add_rows_strided / add_rows_idx with fp32 rows (fp32 gradient stream): exact."""
    R, d, L = 5, 128, 16
    dcls = rnd(128, (R, d)).to(DEV)
    g = torch.full((ops.pad_rows(R * L), d), 3.0, dtype=torch.float32, device=DEV)
    ops.scatter_cls_grad(dcls, g, R, L, R * L)
    ref = torch.zeros(R * L, d, device=DEV)
    ref[::L] = dcls
    assert torch.equal(g[: R * L], ref)
    idx = torch.tensor([0, 7, 19, 30, 41], dtype=torch.int32, device=DEV)
    g2 = torch.full((64, d), 3.0, dtype=torch.float32, device=DEV)
    ops.scatter_cls_grad_idx(dcls, g2, idx, 50)
    ref2 = torch.zeros(50, d, device=DEV)
    ref2[idx.long()] = dcls
    assert torch.equal(g2[:50], ref2)
    add = rnd(129, (R, d)).to(DEV)
    ops.add_rows_strided(g, add, R, L)
    ref[::L] += add
    assert torch.equal(g[: R * L], ref)
    ops.add_rows_idx(g2, add, idx, R)
    ref2[idx.long()] += add
    assert torch.equal(g2[:50], ref2)


@pytest.mark.parametrize("M,N,K", [(4096, 768, 3072), (2048, 768, 2304), (240, 768, 768), (256, 768, 3072)])
def test_gemm_fp32_residual_in_fp32_sum_out_without_bias(M, N, K):
    """The data-gradient flavour of the fp32 gradient stream: out fp32 = A @ B^T + fp32 residual (ring kernel and the small-M kernel)."""
    A, B = bf(rnd(130, (M, K))), bf(rnd(131, (N, K), 0.05))
    res = rnd(132, (M, N))
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
    ops.gemm_nt(A.to(DEV), B.to(DEV), out, M, residual=res.to(DEV))
    ref = A.double() @ B.double().T + res.double()
    close(out, ref, 1e-4, 1e-4 * ref.abs().max().item(), "gemm res32 -> f32")
    out2 = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
    ops.gemm_nt(A.to(DEV), B.to(DEV), out2, M)                                # no residual: EPI_F32
    close(out2, A.double() @ B.double().T, 1e-4, 1e-4 * ref.abs().max().item(), "gemm -> f32")


@pytest.mark.parametrize("accumulate", [False, True])
def test_layernorm_bwd_deferred_group_reduction_is_bit_identical(accumulate):
    """Parameter gradients parked in per-call scratch buffers and reduced by ONE cldrd_ln_reduce_group launch (hip_ops.LnReduceQueue)
    equal the immediate form bit for bit: different row counts, fp32 and bf16 inputs, absent outputs, with and without accumulation."""
    d = 768
    q = ops.LnReduceQueue()
    want, got = [], []
    for j, (T, x32, nones) in enumerate([(4096, True, ()), (240, False, ()), (37, True, (2,)), (8193, False, (0, 1))]):
        x = rnd(60 + j, (T, d), 2.0)
        x = (x if x32 else bf(x)).to(DEV)
        dy = bf(rnd(70 + j, (T, d))).to(DEV)
        gamma = (1 + rnd(80 + j, (d,), 0.1)).to(DEV)
        mean = x.float().mean(1)
        rstd = 1.0 / torch.sqrt(x.float().var(1, unbiased=False) + 1e-12)
        outs = []
        for form in range(2):
            g3 = [None if k in nones else (torch.full((d,), 0.25 * (k + 1), device=DEV) if accumulate else torch.empty(d, device=DEV)) for k in range(3)]
            dx, dx2 = torch.empty(T, d, dtype=torch.bfloat16, device=DEV), torch.empty(T, d, dtype=torch.bfloat16, device=DEV)
            partial = torch.empty(ops.ln_partial_elems(T, d), device=DEV)
            ops.layernorm_bwd(dy, x, mean, rstd, gamma, dx, dx2, g3[0], g3[1], g3[2], partial, T, dropout_p=0.1, seed=5 + j,
                              accumulate=accumulate, defer=q if form else None)
            outs.append((g3, dx, dx2))
        assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
        want.append(outs[0][0]); got.append(outs[1][0])
    assert len(q) == 4
    q.flush(accumulate=accumulate)
    assert len(q) == 0
    for a3, b3 in zip(want, got):
        for a, b in zip(a3, b3):
            assert (a is None and b is None) or torch.equal(a, b)


@pytest.mark.parametrize("V,P,d,M,L", [(300, 40, 256, 6, 20), (2000, 128, 768, 5, 128), (500, 64, 1024, 3, 50)])
def test_embed_ln_fwd_bwd(V, P, d, M, L):
    T = M * L
    word, pos, typ = rnd(20, (V, d)), rnd(21, (P, d)), rnd(22, (2, d))
    gamma, beta = 1 + rnd(23, (d,), 0.1), rnd(24, (d,), 0.1)
    ids = torch.from_numpy(syn.randint(25, 0, V, T))
    dy = bf(rnd(26, (T, d)))
    w, p_, t_ = (a.double().requires_grad_(True) for a in (word, pos, typ))
    g_, b_ = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    e = w[ids] + p_[torch.arange(T) % L] + t_[0]
    y = torch.nn.functional.layer_norm(e, (d,), g_, b_, 1e-12)
    y.backward(dy.double())
    out = torch.empty(T, d, dtype=torch.bfloat16, device=DEV)
    mean, rstd = torch.empty(T, device=DEV), torch.empty(T, device=DEV)
    wd, pd, td, gd, bd = (a.to(DEV) for a in (word, pos, typ, gamma, beta))
    ops.embed_ln_fwd(ids.to(DEV), wd, pd, td[0], gd, bd, out, mean, rstd, T, L, 1e-12)
    close(out, y, 1 / 128, 1e-2, "embed fwd")
    dword, dpos, dtyp = torch.zeros_like(wd), torch.zeros_like(pd), torch.zeros_like(td)
    dg, db = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    partial = torch.empty(ops.ln_partial_elems(T, d), device=DEV)
    ops.embed_ln_bwd(dy.to(DEV), ids.to(DEV), wd, pd, td[0], gd, mean, rstd, dword, dpos, dtyp[0], dg, db, partial, T, L)
    for got, ref, name in ((dword, w.grad, "dword"), (dpos, p_.grad, "dpos"), (dtyp, t_.grad, "dtype"), (dg, g_.grad, "dgamma"),
                           (db, b_.grad, "dbeta")):
        close(got, ref, 1e-3, 1e-3 * ref.abs().max().item(), name)
    # the same from an fp32 dy (fp32 gradient stream): identical values here (dy is bf16-representable), so identical sums up to the atomics' order
    dword2, dpos2, dtyp2 = torch.zeros_like(wd), torch.zeros_like(pd), torch.zeros_like(td)
    dg2, db2 = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    ops.embed_ln_bwd(dy.float().to(DEV), ids.to(DEV), wd, pd, td[0], gd, mean, rstd, dword2, dpos2, dtyp2[0], dg2, db2, partial, T, L)
    for got, ref, name in ((dword2, dword, "dword32"), (dpos2, dpos, "dpos32"), (dtyp2, dtyp, "dtype32"), (dg2, dg, "dgamma32"), (db2, db, "dbeta32")):
        close(got, ref, 1e-5, 1e-5 * ref.abs().max().item(), name)
    # ... and from an fp32 stream + a bf16 branch term that add up to the same dy
    half = bf(dy.float() * 0.5)
    dword3, dpos3, dtyp3 = torch.zeros_like(wd), torch.zeros_like(pd), torch.zeros_like(td)
    dg3, db3 = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    ops.embed_ln_bwd((dy.float() - half.float()).to(DEV), ids.to(DEV), wd, pd, td[0], gd, mean, rstd, dword3, dpos3, dtyp3[0], dg3, db3, partial, T, L,
                     dy_branch=half.to(DEV))
    for got, ref, name in ((dword3, dword, "dword (stream + branch)"), (dpos3, dpos, "dpos"), (dg3, dg, "dgamma"), (db3, db, "dbeta")):
        close(got, ref, 1e-4, 1e-5 * ref.abs().max().item(), name)


def test_colsum_and_scatter():
    T, N = 1000, 2304
    x = bf(rnd(27, (T, N)))
    out = torch.zeros(N, device=DEV)
    partial = torch.empty(((T + 127) // 128) * N, device=DEV)
    ops.colsum(x.to(DEV), out, partial, T, accumulate=False)
    close(out, x.double().sum(0), 1e-4, 1e-3, "colsum")
    R, d, L = 5, 128, 16
    dcls = rnd(28, (R, d))
    g = torch.full((ops.pad_rows(R * L), d), 3.0, dtype=torch.bfloat16, device=DEV)
    ops.scatter_cls_grad(dcls.to(DEV), g, R, L, R * L)
    ref = torch.zeros(R * L, d)
    ref[::L] = dcls
    close(g[: R * L], ref, 1 / 256, 0, "scatter_cls")


# ------------------------------------------------------------------------------------------------ scoring / losses
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_score_fwd_bwd(mode):
    from oracle.encoder_ref import in_batch_index
    B, N, d = 4, 6, 128
    q, p = rnd(30, (B, d)), rnd(31, (B * N, d))
    idx = torch.arange(B)[:, None] * N + torch.arange(N)[None] if mode == 0 else in_batch_index(B, N, mode == 1)
    qr, pr = q.double().requires_grad_(True), p.double().requires_grad_(True)
    ref = (qr[:, None, :] * pr[idx]).sum(-1)
    dl = rnd(32, tuple(ref.shape))
    ref.backward(dl.double())
    logits = torch.empty(tuple(ref.shape), dtype=torch.float32, device=DEV)
    ops.score_fwd(q.to(DEV), p.to(DEV), logits, B, N, mode)
    close(logits, ref, 1e-5, 1e-4, "score fwd")
    dq, dp = torch.empty(B, d, device=DEV), torch.empty(B * N, d, device=DEV)
    ops.score_bwd(dl.to(DEV), q.to(DEV), p.to(DEV), dq, dp, B, N, mode)
    close(dq, qr.grad, 1e-5, 1e-4, "score dq")
    close(dp, pr.grad, 1e-5, 1e-4, "score dp")


G = np.load(os.path.join(os.path.dirname(__file__), "golden", "losses.npz"))


@pytest.mark.parametrize("name", [str(n) for n in G["names"]])
def test_losses_match_reference_goldens(name):
    """HIP loss kernels against the vectors captured from the reference losses/*.py (value and gradient)."""
    from cldrd_amd import losses as Lh
    kind = str(G[name + "/kind"])
    yp = torch.tensor(G[name + "/y_pred"], device=DEV, requires_grad=True)
    yt = torch.tensor(G[name + "/y_true"], device=DEV)
    kw = {k.split("/kw_")[1]: G[k] for k in G.files if k.startswith(name + "/kw_")}
    red = str(kw["reduction"]) if "reduction" in kw else "mean"
    if kind == "kl":
        loss = Lh.KLDiv(float(kw["T"]))(yp, yt)
    elif kind == "mse":
        loss = Lh.MarginMSE()(yp, yt)
    elif kind == "ranknet":
        loss = Lh.ranknet_loss(yp, yt, reduction=red)
    elif kind == "lambda":
        loss = Lh.lambda_mrr_loss(yp, yt, reduction=red)
    else:
        loss = Lh.bweight_lambda_mrr_loss(yp, yt, torch.tensor(kw["batch_weight"], device=DEV), reduction=red)
    loss.backward()
    ref_v, ref_g = float(G[name + "/value"]), G[name + "/grad"]
    assert loss.item() == pytest.approx(ref_v, rel=3e-5, abs=1e-6)          # both sides fp32
    scale = float(np.abs(ref_g).max())
    assert np.allclose(yp.grad.cpu().numpy(), ref_g, rtol=5e-4, atol=3e-5 * scale + 1e-9)


def test_loss_error_behaviour():
    from cldrd_amd import losses as Lh
    yp = torch.zeros(2, 3, device=DEV)
    yt = torch.tensor([[1.0, 0.5, -1.0], [1.0, 0.5, 0.0]], device=DEV)
    with pytest.raises(ValueError):
        Lh.lambda_mrr_loss(yp, yt, reduction="max")
    with pytest.raises(AssertionError):
        Lh.ranknet_loss(yp, yt)                                               # -1 labels asserted absent (ranknet.py:16)
    with pytest.raises(AssertionError):
        Lh.KLDiv()(torch.zeros(3, device=DEV), torch.zeros(3, device=DEV))   # inputs must be 2-D (kl_div.py:17)
    before = yp.clone()
    Lh.lambda_mrr_loss(yp, yt)
    assert torch.equal(yp, before)                                            # inputs are not mutated


# ------------------------------------------------------------------------------------------------ optimizer
def test_clip_and_adamw_match_oracle():
    from oracle import optim_ref as O
    n = 64 * 1000
    p, g = rnd(40, (n,)), rnd(41, (n,), 0.01)
    m, v = rnd(42, (n,), 0.001), rnd(43, (n,), 0.001).abs() * 1e-3
    flags = torch.zeros(n // 64, dtype=torch.uint8)
    flags[::2] = 1
    pd, gd, md, vd = (a.to(DEV).clone() for a in (p, g, m, v))
    clip = torch.zeros(3, device=DEV)
    partial = torch.empty(ops.sqnorm_blocks(), device=DEV)
    ops.grad_clip_coef(gd, 1.0, partial, clip)
    total, coef = O.clip_coef([g.numpy()], 1.0)
    assert clip[0].item() == pytest.approx(total, rel=1e-5) and clip[1].item() == pytest.approx(coef, rel=1e-5)
    assert clip[2].item() == 0.0
    shadow = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    ops.adamw_step(pd, gd, md, vd, flags.to(DEV), shadow, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, step=3,
                   clip=clip)
    dec = flags.repeat_interleave(64).bool().numpy()
    gn = g.double().numpy() * coef
    p1, m1, v1 = O.adamw_step(p.numpy(), gn, m.numpy(), v.numpy(), lr=1e-3, step=3, weight_decay=0.0)
    p2, _, _ = O.adamw_step(p.numpy(), gn, m.numpy(), v.numpy(), lr=1e-3, step=3, weight_decay=0.01)
    ref_p = np.where(dec, p2, p1)
    assert np.allclose(pd.cpu().numpy(), ref_p, rtol=1e-5, atol=1e-6)
    assert np.allclose(md.cpu().numpy(), m1, rtol=1e-5, atol=1e-8)
    assert np.allclose(vd.cpu().numpy(), v1, rtol=1e-5, atol=1e-10)
    assert torch.equal(shadow.cpu(), pd.cpu().to(torch.bfloat16))
    # non-finite gradients leave the weights untouched
    gd[5] = float("inf")
    ops.grad_clip_coef(gd, 1.0, partial, clip)
    before = pd.clone()
    ops.adamw_step(pd, gd, md, vd, flags.to(DEV), None, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, step=4, clip=clip)
    assert clip[2].item() == 1.0 and torch.equal(pd, before)


def test_clip_norm_taken_in_two_pieces_matches_the_one_piece_norm_and_the_oracle():
    """cldrd_sqnorm_partial over [0, split) and [split, n) into the two halves of the partial buffer + cldrd_clip_coef = the norm /
    coefficient of cldrd_grad_clip_coef over the whole vector (the trainer takes the first piece under the last weight-gradient launch)."""
    from oracle import optim_ref as O
    n, split = 64 * 5000, 64 * 1237
    g = rnd(44, (n,), 0.01)
    gd = g.to(DEV)
    nb = ops.sqnorm_blocks()
    partial = torch.full((nb,), float("nan"), device=DEV)
    one, two = torch.zeros(3, device=DEV), torch.zeros(3, device=DEV)
    ops.grad_clip_coef(gd, 1.0, torch.empty(nb, device=DEV), one)
    ops.sqnorm_partial(gd[:split], partial, nb // 2)
    ops.sqnorm_partial(gd[split:], partial[nb // 2:], nb // 2)
    ops.clip_coef(partial, nb, 1.0, two)
    total, coef = O.clip_coef([g.numpy()], 1.0)
    assert two[0].item() == pytest.approx(total, rel=1e-6) and two[1].item() == pytest.approx(coef, rel=1e-6) and two[2].item() == 0.0
    assert two[0].item() == pytest.approx(one[0].item(), rel=1e-6)
    gd[split + 3] = float("nan")
    ops.sqnorm_partial(gd[split:], partial[nb // 2:], nb // 2)
    ops.clip_coef(partial, nb, 1.0, two)
    assert two[2].item() == 1.0


def test_copy_segments_copies_up_to_eight_buffers_in_one_launch():
    srcs = [torch.arange(8 * 32, device=DEV), torch.arange(256 * 128, device=DEV).view(256, 128) * 3,
            (torch.arange(256 * 128, device=DEV).view(256, 128) % 2), torch.linspace(-1, 1, 8 * 32, device=DEV).view(8, 32),
            torch.arange(7, device=DEV, dtype=torch.uint8)]                  # the last one: 7 bytes, the byte-wise path
    dsts = [torch.zeros_like(t) for t in srcs]
    ops.copy_segments(dsts, srcs)
    for d, t in zip(dsts, srcs):
        assert torch.equal(d, t)
    with pytest.raises(ValueError):
        ops.copy_segments(dsts[:1], [srcs[0].float()])
    with pytest.raises(ValueError):
        ops.copy_segments([], [])


def test_zero_segments_clears_exactly_the_given_ranges():
    """cldrd_zero_segments (the trainer's zero_grad: the embedding-table gradient ranges of both towers in one launch): the ranges are zero,
    their neighbours untouched; unaligned ranges are refused."""
    buf = torch.full((3_000_000,), 7.0, device=DEV)
    a, b, c = buf[64:64 + 1_000_000], buf[1_500_032:1_500_032 + 400_000], buf[2_999_996:3_000_000]
    ops.zero_segments([a, b, c])
    keep = torch.ones_like(buf, dtype=torch.bool)
    keep[64:64 + 1_000_000] = False
    keep[1_500_032:1_500_032 + 400_000] = False
    keep[2_999_996:] = False
    assert not buf[~keep].any() and (buf[keep] == 7.0).all()
    with pytest.raises(Exception):
        ops.zero_segments([buf[1:5]])                 # 4-byte offset: not 16-byte aligned
    with pytest.raises(ValueError):
        ops.zero_segments([])


def test_adamw_step_writes_the_fp16_shadow_of_a_sub_range():
    """The fused step leaves the fp16 copy of parameters [lo, hi) (what cast_f16 of the updated parameters gives) and changes
    nothing else: p, m, v and the bf16 shadow are bit-identical to the step without it; a skipped step still refreshes the copy."""
    n, lo, hi = 64 * 500, 64 * 100 + 4, 64 * 300 - 8
    p, g = rnd(50, (n,)), rnd(51, (n,), 0.01)
    m, v = rnd(52, (n,), 0.001), rnd(53, (n,), 0.001).abs() * 1e-3
    flags = torch.zeros(n // 64, dtype=torch.uint8); flags[1::3] = 1
    res = []
    for fused in (False, True):
        pd, gd, md, vd = (a.to(DEV).clone() for a in (p, g, m, v))
        shadow = torch.empty(n, dtype=torch.bfloat16, device=DEV)
        h16 = torch.full((hi - lo + 8,), 7.0, dtype=torch.float16, device=DEV)
        ops.adamw_step(pd, gd, md, vd, flags.to(DEV), shadow, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, step=2,
                       shadow16=h16 if fused else None, h16_range=(lo, hi) if fused else None)
        res.append((pd, md, vd, shadow, h16))
    for a, b in zip(res[0][:4], res[1][:4]):
        assert torch.equal(a, b)
    h16 = res[1][4]
    assert torch.equal(h16[:hi - lo], res[1][0][lo:hi].half()) and (h16[hi - lo:] == 7.0).all()
    with pytest.raises(Exception):
        ops.adamw_step(pd, gd, md, vd, flags.to(DEV), shadow, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, step=2,
                       shadow16=h16, h16_range=(lo + 1, hi))


def test_adamw_flag_bit_1_leaves_the_shadows_of_a_chunk_unwritten():
    """decay_flags bit 1 (embedding tables: read in fp32, nobody reads their 16-bit shadows): p / m / v are updated exactly as without
    the bit, the bf16 and fp16 shadows of those 64-element chunks keep their old contents."""
    n = 64 * 40
    p, g = rnd(60, (n,)), rnd(61, (n,), 0.01)
    m, v = rnd(62, (n,), 0.001), rnd(63, (n,), 0.001).abs() * 1e-3
    base = torch.zeros(n // 64, dtype=torch.uint8); base[::2] = 1
    skip = base.clone(); skip[5:17] |= 2
    outs = []
    for flags in (base, skip):
        pd, gd, md, vd = (a.to(DEV).clone() for a in (p, g, m, v))
        shadow = torch.full((n,), 3.0, dtype=torch.bfloat16, device=DEV)
        h16 = torch.full((n,), 5.0, dtype=torch.float16, device=DEV)
        ops.adamw_step(pd, gd, md, vd, flags.to(DEV), shadow, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, step=2,
                       shadow16=h16, h16_range=(0, n))
        outs.append((pd, md, vd, shadow, h16))
    for a, b in zip(outs[0][:3], outs[1][:3]):
        assert torch.equal(a, b)
    lo, hi = 5 * 64, 17 * 64
    assert (outs[1][3][lo:hi] == 3.0).all() and (outs[1][4][lo:hi] == 5.0).all()
    keep = torch.ones(n, dtype=torch.bool, device=DEV); keep[lo:hi] = False
    assert torch.equal(outs[1][3][keep], outs[0][3][keep]) and torch.equal(outs[1][4][keep], outs[0][4][keep])


def test_transpose_cast_batched():
    src = rnd(44, (128 * 96 + 64 * 200,))
    desc = torch.tensor([0, 0, 128, 96, 128 * 96, 128 * 96, 64, 200], dtype=torch.int64, device=DEV)
    t0, t1 = 4 * 3, 2 * 7
    prefix = torch.tensor([0, t0, t0 + t1], dtype=torch.int32, device=DEV)
    dst = torch.zeros(src.numel(), dtype=torch.bfloat16, device=DEV)
    ops.transpose_cast_batched(src.to(DEV), dst, desc, prefix, 2, t0 + t1)
    a = src[: 128 * 96].view(128, 96).T.contiguous().to(torch.bfloat16)
    b = src[128 * 96:].view(64, 200).T.contiguous().to(torch.bfloat16)
    assert torch.equal(dst[: 128 * 96].cpu().view(96, 128), a) and torch.equal(dst[128 * 96:].cpu().view(200, 64), b)


# ------------------------------------------------------------------------------------------------ dropout masks (exact)
from oracle import dropout_ref as DR


@pytest.mark.parametrize("M,N,K,seed", [(300, 256, 64, 1234), (2048, 768, 128, (7 << 32) | 99), (1100, 384, 64, 5)])
def test_gemm_dropout_mask_is_the_oracle_mask(M, N, K, seed):
    """out = dropout(A.B^T) + residual with the mask of oracle/dropout_ref.py at (row, col) = (m, n), scale 1/(1-p)."""
    p = 0.1
    A, B = bf(rnd(40, (M, K))), bf(rnd(41, (N, K)))
    res = bf(rnd(42, (M, N)))
    plain = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm_nt(A.to(DEV), B.to(DEV), plain)
    out = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm_nt(A.to(DEV), B.to(DEV), out, residual=res.to(DEV), dropout_p=p, seed=seed)
    keep = torch.from_numpy(DR.keep_mask(seed, p, M, N))
    ref = torch.where(keep, plain.cpu() / (1 - p), torch.zeros(())) + res.float()
    close(out, ref, 1e-6, 1e-6, "gemm dropout mask")
    assert abs(keep.float().mean().item() - (1 - p)) < 0.01


@pytest.mark.parametrize("T,d", [(513, 768), (64, 256)])
def test_layernorm_bwd_dropout_branch_uses_the_oracle_mask(T, d):
    p, seed = 0.1, 31337
    x, dy = bf(rnd(43, (T, d), 2.0)), bf(rnd(44, (T, d)))
    gamma = 1 + rnd(45, (d,), 0.1)
    mean = x.float().mean(1)
    rstd = 1.0 / torch.sqrt(x.float().var(1, unbiased=False) + 1e-12)
    dx = torch.empty(T, d, dtype=torch.bfloat16, device=DEV)
    dx2 = torch.empty_like(dx)
    dg, db, dbias = (torch.zeros(d, device=DEV) for _ in range(3))
    partial = torch.empty(ops._lib.load().cldrd_ln_partial_blocks(T) * 3 * d, device=DEV)
    ops.layernorm_bwd(dy.to(DEV), x.to(DEV), mean.to(DEV), rstd.to(DEV), gamma.to(DEV), dx, dx2, dg, db, dbias, partial, T,
                      dropout_p=p, seed=seed, accumulate=False)
    keep = torch.from_numpy(DR.keep_mask(seed, p, T, d))
    # dx2 is computed from the unrounded dx: compare against the fp32 formula, one bf16 rounding
    xh = (x.float() - mean[:, None]) * rstd[:, None]
    t = dy.float() * gamma
    ref = rstd[:, None] * (t - t.mean(1, keepdim=True) - xh * (t * xh).mean(1, keepdim=True))
    close(dx, ref, 1 / 128, 1e-3, "ln_bwd dx")
    close(dx2, torch.where(keep, ref / (1 - p), torch.zeros(())), 1 / 128, 1e-3, "ln_bwd dropped dx")
    assert torch.equal(dx2.cpu() == 0, ~keep | (dx2.cpu() == 0))
    close(dbias, torch.where(keep, ref / (1 - p), torch.zeros(())).sum(0), 1e-3, 1e-2, "bias gradient of the dropped branch")


@pytest.mark.parametrize("nseq,L,H", [(2, 64, 2), (3, 128, 1), (2, 30, 2), (1, 256, 2), (2, 192, 1), (1, 130, 1)])
def test_attention_dropout_fwd_bwd_against_the_oracle_mask(nseq, L, H):
    """softmax -> dropout(mask of oracle/dropout_ref.py, scale 1/(1-p)) -> . V, forward and both backward sweeps."""
    p, seed = 0.25, 4242
    d, T = H * 64, nseq * L
    qkv = bf(rnd(46, (T, 3 * d), 1.0))
    keep = torch.from_numpy(DR.attention_keep_mask(seed, p, nseq, H, L))
    qv = qkv.double().requires_grad_(True)
    x = qv.view(nseq, L, 3, H, 64)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    P = torch.softmax(q @ k.transpose(2, 3) * 0.125, -1)
    ref = ((P * keep / (1 - p)) @ v).transpose(1, 2).reshape(T, d)
    dctx = bf(rnd(47, (T, d)))
    ref.backward(dctx.double())
    ctx = torch.empty(T, d, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(nseq, H, L, dtype=torch.float32, device=DEV)
    ops.attention_fwd(qkv.to(DEV), None, ctx, lse, nseq, L, H, dropout_p=p, seed=seed)
    close(ctx, ref, 1 / 64, 3e-2, "attention fwd with dropout")
    dqkv = torch.zeros(T, 3 * d, dtype=torch.bfloat16, device=DEV)
    ops.attention_bwd(qkv.to(DEV), None, ctx, dctx.to(DEV), lse, dqkv, nseq, L, H, dropout_p=p, seed=seed)
    g = qv.grad.float()
    close(dqkv, g, 1 / 32, 2e-2 * g.abs().max().item(), "attention bwd with dropout")


# ------------------------------------------------------------------------------------------------ lambda_loss / weighted pointwise
import json as _json

from oracle import losses_ref as LR

_G2 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "losses2.npz"))
_META2 = _json.loads(str(_G2["meta"]))


@pytest.mark.parametrize("name", sorted(_META2, key=lambda s: int(s[4:])))
def test_lambda_loss_kernel_against_reference_goldens(name):
    """fp32 kernel vs the reference's float64 autograd (tests/golden/losses2.npz): 2e-4 relative on the value, 2e-4 of the largest
    gradient entry on the gradient (exp2 / pow / log in fp32)."""
    from cldrd_amd.losses import lambda_loss
    kw = _META2[name]
    yp = torch.tensor(_G2["y_pred"], dtype=torch.float32, device=DEV, requires_grad=True)
    yt = torch.tensor(_G2["y_true"], dtype=torch.float32, device=DEV)
    out = lambda_loss(yp, yt, **kw)
    out.backward()
    ref_v, ref_g = float(_G2[name + ".value"]), _G2[name + ".grad"]
    assert abs(out.item() - ref_v) <= 2e-4 * abs(ref_v) + 1e-6, (kw, out.item(), ref_v)
    assert np.abs(yp.grad.cpu().numpy() - ref_g).max() <= 2e-4 * np.abs(ref_g).max() + 1e-7, kw


def test_lambda_loss_demo_and_bigger_slate_vs_oracle():
    from cldrd_amd.losses import lambda_loss
    for t, printed in (("demo.t1", 0.0127), ("demo.t2", 0.0110)):
        yp = torch.tensor(_G2["demo.y_pred"], device=DEV)
        out = lambda_loss(yp, torch.tensor(_G2[t], device=DEV), weighing_scheme="ndcgLoss1_scheme", reduction_log="natural")
        assert round(out.item(), 4) == printed
    B, N = 8, 200                                              # cfg3-sized slates, against the float64 oracle
    yp = rnd(50, (B, N), 3.0)
    yt = torch.from_numpy(syn.labels_mode9(B, 30)).repeat(1, 7)[:, :N].contiguous()
    yt[:, 190:] = -1.0
    for sch in (None, "ndcgLoss2PP_scheme", "lambdaRank_scheme", "rankNetWeightedByGTDiffPowed_scheme"):
        ypd = yp.to(DEV).requires_grad_(True)
        out = lambda_loss(ypd, yt.to(DEV), weighing_scheme=sch, k=50)
        out.backward()
        v, g = LR.lambda_loss(yp.numpy(), yt.numpy(), weighing_scheme=sch, k=50)
        assert abs(out.item() - v) <= 2e-4 * abs(v) + 1e-6
        assert np.abs(ypd.grad.cpu().numpy() - g).max() <= 5e-4 * np.abs(g).max() + 1e-8


def test_weighted_pointwise_kernel():
    from cldrd_amd.losses import weighted_pointwise_loss
    for i in (0, 1):
        yp = torch.tensor(_G2[f"wp.demo{i}.pred"], dtype=torch.float32, device=DEV, requires_grad=True)
        out = weighted_pointwise_loss(yp, torch.tensor(_G2["wp.weight"], device=DEV))
        out.backward()
        assert abs(out.item() - float(_G2[f"wp.demo{i}.value"])) < 1e-6
        assert np.allclose(yp.grad.cpu().numpy(), _G2[f"wp.demo{i}.grad"], rtol=1e-4, atol=1e-8)
    yp = torch.tensor(_G2["y_pred"], dtype=torch.float32, device=DEV, requires_grad=True)
    out = weighted_pointwise_loss(yp, torch.tensor(_G2["wp.rand.weight"], dtype=torch.float32, device=DEV), T=0.7)
    out.backward()
    assert abs(out.item() - float(_G2["wp.rand.value"])) < 1e-5 and np.allclose(yp.grad.cpu().numpy(), _G2["wp.rand.grad"], rtol=1e-4, atol=1e-8)
    with pytest.raises(AssertionError):
        weighted_pointwise_loss(yp, -torch.ones_like(yp))


def test_transpose_bf16_batched_matches_torch():
    shapes = [(128, 64), (768, 2304), (192, 3072)]
    total = sum(r * c for r, c in shapes)
    src = bf(rnd(60, (total,))).to(DEV)
    dst = torch.zeros(total, dtype=torch.bfloat16, device=DEV)
    desc, prefix, off, tiles = [], [0], 0, 0
    for r, c in shapes:
        desc += [off, off, r, c]
        off += r * c
        tiles += (r // 64) * (c // 64)
        prefix.append(tiles)
    ops.transpose_bf16_batched(src, dst, torch.tensor(desc, dtype=torch.int64, device=DEV), torch.tensor(prefix, dtype=torch.int32, device=DEV),
                               len(shapes), tiles)
    off = 0
    for r, c in shapes:
        assert torch.equal(dst[off:off + r * c].view(c, r), src[off:off + r * c].view(r, c).T), (r, c)
        off += r * c


# ------------------------------------------------------------------------------------------------ fp16 forward format
def test_forward_kernels_in_the_fp16_format():
    """The high-precision forward of the query tower runs the same kernels on fp16 operands (encoder.py): small-M GEMM with the
    forward epilogues, attention (all-scores-in-registers kernel and the CLS-only one), LayerNorm / embedding outputs.  fp16 keeps
    11 significant bits: the tolerances are 8x tighter than for the bf16 format."""
    M, N, K = 240, 768, 768
    A, B = rnd(41, (M, K)).half(), rnd(42, (N, K), 0.05).half()
    bias, res32 = rnd(43, (N,)), rnd(44, (M, N))
    ref = A.double() @ B.double().T + bias.double()
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ops.gemm_nt(A.to(DEV), B.to(DEV), out, bias=bias.to(DEV))
    close(out, ref, 1.0 / 1024, 1e-3, "fp16 gemm + bias")
    ops.gemm_nt(A.to(DEV), B.to(DEV), out, bias=bias.to(DEV), act=1)
    close(out, torch.nn.functional.gelu(ref), 1.0 / 1024, 1e-3, "fp16 gemm + bias + gelu")
    out32 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm_nt(A.to(DEV), B.to(DEV), out32, bias=bias.to(DEV), residual=res32.to(DEV))
    close(out32, ref + res32.double(), 1e-4, 2e-4 * math.sqrt(K), "fp16 gemm + bias + fp32 residual -> fp32")
    # a tape in this format (round 4, the all-fp16 training mode: tests/test_gpu_amp16.py) takes fp16 `preact`; a bf16 one is refused here
    with pytest.raises(Exception):
        ops.gemm_nt(A.to(DEV), B.to(DEV), out, bias=bias.to(DEV), gelu_pre=torch.empty(M, N, dtype=torch.bfloat16, device=DEV), act=2)
    # attention
    for nseq, L, H in ((3, 30, 2), (2, 128, 3)):
        d, T = H * 64, nseq * L
        qkv = rnd(45, (T, 3 * d)).half()
        lens = np.clip(syn.msmarco_lengths(46, nseq, L), 2, L)
        mask = torch.from_numpy((np.arange(L)[None, :] < lens[:, None]).astype(np.int64))
        ref_ctx, ref_lse = attn_ref(qkv.double(), mask, nseq, L, H)
        ctx = torch.empty(T, d, dtype=torch.float16, device=DEV)
        lse = torch.empty(nseq, H, L, dtype=torch.float32, device=DEV)
        ops.attention_fwd(qkv.to(DEV), mask.to(DEV), ctx, lse, nseq, L, H)
        close(ctx, ref_ctx, 1 / 512, 3e-3, f"fp16 attention L={L}")
        close(lse, ref_lse, 1e-4, 1e-3, "fp16 attention lse")
        # CLS-only form: queries of token 0 against K | V of every token
        kv = qkv[:, d:].contiguous()
        qc = qkv.view(nseq, L, 3 * d)[:, 0, :d].contiguous()
        ctxc = torch.empty(nseq, d, dtype=torch.float16, device=DEV)
        probs = torch.empty(nseq, H, L, dtype=torch.float32, device=DEV)
        ops.attention_cls_fwd(qc.to(DEV), kv.to(DEV), mask.to(DEV), ctxc, probs, nseq, L, H)
        close(ctxc, ref_ctx.view(nseq, L, d)[:, 0], 1 / 1024, 2e-3, f"fp16 CLS attention L={L}")
    # L > 128 in fp16 goes through the whole kernel family of the bf16 path since round 4 (io_f16 = 5)
    qkv = rnd(50, (2 * 256, 192)).half()
    ref_ctx, _ = attn_ref(qkv.double(), torch.ones(2, 256, dtype=torch.int64), 2, 256, 1)
    ctx = torch.empty(512, 64, dtype=torch.float16, device=DEV)
    ops.attention_fwd(qkv.to(DEV), None, ctx, None, 2, 256, 1)
    close(ctx, ref_ctx, 1 / 512, 3e-3, "fp16 attention L=256")
    # LayerNorm: fp32 sum in, fp16 + fp32 out
    T, d = 77, 768
    x = rnd(47, (T, d), 2.0)
    g_, b_ = 1.0 + rnd(48, (d,), 0.1), rnd(49, (d,), 0.1)
    ref_ln = torch.nn.functional.layer_norm(x.double(), (d,), g_.double(), b_.double(), 1e-12)
    o16 = torch.empty(ops.pad_rows(T), d, dtype=torch.float16, device=DEV)
    o32 = torch.empty(ops.pad_rows(T), d, dtype=torch.float32, device=DEV)
    xs = torch.zeros(ops.pad_rows(T), d)
    xs[:T] = x
    ops.layernorm_fwd(xs.to(DEV), g_.to(DEV), b_.to(DEV), o16, None, None, T, 1e-12, out32=o32)
    close(o16[:T], ref_ln, 1 / 1024, 1e-3, "LayerNorm fp16 out")
    close(o32[:T], ref_ln, 1e-5, 1e-5, "LayerNorm fp32 out")


def test_torch_library_ops_on_the_gpu():
    """torch.ops.cldrd.* run the HIP kernels, carry autograd formulas, and pass torch.library.opcheck (schema, fake kernel,
    autograd registration consistent with the real kernel)."""
    import cldrd_amd.torch_ops  # noqa: F401
    q = torch.randn(4, 768, device=DEV, requires_grad=True)
    p = torch.randn(32, 768, device=DEV, requires_grad=True)
    for mode in (0, 1, 2):
        logits = torch.ops.cldrd.nway_score(q, p, 4, 8, mode)
        ref = (q.detach().double() @ p.detach().double().T)
        if mode == 0:
            ref = torch.stack([ref[b, b * 8:(b + 1) * 8] for b in range(4)])
            close(logits, ref, 1e-5, 1e-4, "nway_score")
        logits.sum().backward()
    torch.library.opcheck(torch.ops.cldrd.nway_score, (q.detach(), p.detach(), 4, 8, 0), test_utils=("test_schema", "test_faketensor"))
    # dense but non-contiguous inputs (transposed views): the gradients must come back in the inputs' logical layout
    qt = torch.randn(768, 4, device=DEV).t().requires_grad_(True)
    pt = torch.randn(768, 32, device=DEV).t().requires_grad_(True)
    wgt = torch.randn(4, 8, device=DEV)
    (torch.ops.cldrd.nway_score(qt, pt, 4, 8, 0) * wgt).sum().backward()
    qc, pc = qt.detach().contiguous().requires_grad_(True), pt.detach().contiguous().requires_grad_(True)
    (torch.ops.cldrd.nway_score(qc, pc, 4, 8, 0) * wgt).sum().backward()
    assert torch.equal(qt.grad, qc.grad) and torch.equal(pt.grad, pc.grad)
    yp = torch.randn(4, 8, device=DEV, requires_grad=True)
    yt = torch.randn(4, 8, device=DEV)
    out, grad = torch.ops.cldrd.listwise_loss(yp, yt, 1, None, 1.0, -1.0, True)
    out[0].backward()
    assert torch.allclose(yp.grad, grad)
    torch.library.opcheck(torch.ops.cldrd.listwise_loss, (yp.detach(), yt, 1, None, 1.0, -1.0, True), test_utils=("test_schema", "test_faketensor"))
    x = torch.randn(64, 128, device=DEV).bfloat16()
    w = torch.randn(256, 128, device=DEV).bfloat16()
    y = torch.ops.cldrd.linear(x, w, None, None, False, True)
    close(y, x.double() @ w.double().T, 1e-4, 1e-3, "cldrd::linear")
    ln = torch.ops.cldrd.layer_norm(y, torch.ones(256, device=DEV), torch.zeros(256, device=DEV), 1e-12)
    close(ln, torch.nn.functional.layer_norm(y.double(), (256,)), 1 / 128, 1e-2, "cldrd::layer_norm")
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.cldrd.nway_score(q.detach().cpu(), p.detach().cpu(), 4, 8, 0)


@pytest.mark.parametrize("M", [1024, 240])
def test_gelu_epilogue_accuracy_over_every_fp16_input(M):
    """Round 5: GELU in the erfc form (common.h: gelu_q; x Phi(x) = max(x, 0) - |x| erfc(|x| / sqrt 2) / 2, Abramowitz-Stegun 7.1.26).  EVERY
    fp16 value with |x| <= 6 goes through the FFN1 forward epilogue of both GEMM kernels (identity weights: the accumulator IS x): the fp16 result
    is within 1 fp16 ulp of the correctly rounded erf-GELU wherever |gelu(x)| >= 2^-10 and within 2e-7 |x| + 1 subnormal step absolutely below
    that (the far negative tail, where A-S's absolute 7.5e-8 on Phi shows); the saved derivative gelu'(x) within 1 ulp / 1e-6."""
    K = 256
    bits = np.arange(0, 0x4600 + 1, dtype=np.uint16)                 # +0 .. 6.0
    vals = np.concatenate([bits.view(np.float16), (bits | 0x8000).view(np.float16)])
    n = M * K
    reps = int(np.ceil(n / vals.size))
    x = np.tile(vals, reps)[:n].reshape(M, K)
    if M * K < vals.size:                                            # the small-M kernel: sample the values evenly
        x = vals[np.linspace(0, vals.size - 1, n).astype(np.int64)].reshape(M, K)
    xd = torch.from_numpy(x).to(DEV)
    eye = torch.eye(K, dtype=torch.float16, device=DEV)
    h = torch.empty(ops.pad_rows(M), K, dtype=torch.float16, device=DEV)
    dv = torch.empty(ops.pad_rows(M), K, dtype=torch.float16, device=DEV)
    ops.gemm_nt(xd, eye, h, M, bias=torch.zeros(K, device=DEV), preact=dv, act=3)
    h2 = torch.empty_like(h)
    ops.gemm_nt(xd, eye, h2, M, bias=torch.zeros(K, device=DEV), act=1)            # evaluation flavour (no tape)
    assert torch.equal(h[:M], h2[:M])
    from scipy.special import erf
    xf = x.astype(np.float64)
    ref = xf * 0.5 * (1.0 + erf(xf / np.sqrt(2.0)))
    got = h[:M].cpu().numpy().astype(np.float64)
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(ref), 2.0 ** -14))) - 10)          # fp16 spacing at ref (subnormal floor 2^-24)
    err = np.abs(got - ref)
    big = np.abs(ref) >= 2.0 ** -10
    assert (err[big] <= 1.0 * ulp[big]).all(), f"max error {np.max(err[big] / ulp[big]):.3f} ulp"
    assert (err[~big] <= 2e-7 * np.abs(xf[~big]) + 2.0 ** -24 + 0.5 * ulp[~big]).all()
    dref = 0.5 * (1.0 + erf(xf / np.sqrt(2.0))) + xf * np.exp(-0.5 * xf * xf) / np.sqrt(2.0 * np.pi)
    dgot = dv[:M].cpu().numpy().astype(np.float64)
    dulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(dref), 2.0 ** -14))) - 10)
    assert (np.abs(dgot - dref) <= 1.0 * dulp + 1e-6).all()
