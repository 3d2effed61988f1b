"""The all-fp16 training mode (round 4, encoder.py: amp16; the reference trains under fp16 autocast + GradScaler,
trainer/multistep-curriculum/nway_listwise_1.py:334-359): the fp16 instantiations of the backward kernels against fp64 torch-CPU on the
same fp16-rounded inputs (tolerance: fp16's 2^-10 instead of bf16's 2^-7), the loss-scale plumbing, and the mode end to end."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import cldrd_amd.synthetic as syn
import selftest
from cldrd_amd import hip_ops as ops
from cldrd_amd.encoder import EncoderConfig
from cldrd_amd.trainer import NwayTrainer

DEV = "cuda"


def rnd(seed, shape, scale=1.0):
    return torch.from_numpy(syn.normal(seed, int(np.prod(shape))).reshape(shape).astype(np.float32)) * scale


def close(got, ref, rtol, atol, what=""):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    err = (got - ref).abs()
    bad = err > atol + rtol * ref.abs()
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.numel()} off, max err {err.max():.4e} at |ref| {ref.abs().max():.3e}"


@pytest.mark.parametrize("M,N1,N2", [(200, 128, 256), (4096, 768, 768), (3000, 256, 768), (5000, 1536, 768), (130, 256, 192)])
def test_wgrad_fp16_operands_and_inverse_scale(M, N1, N2):
    dY, X = rnd(1, (M, N1)).half(), rnd(2, (M, N2)).half()
    ref = dY.double().T @ X.double()
    refb = dY.double().sum(0)
    dW = torch.empty(N1, N2, device=DEV)
    db = torch.empty(N1, device=DEV)
    ws = torch.empty(max(1, ops.wgrad_workspace_elems(M, N1, N2)), device=DEV)
    ops.wgrad(dY.to(DEV), X.to(DEV), dW, M, ws, dbias=db)
    scale = ref.abs().max().item()
    close(dW, ref, 1e-3, 2e-4 * scale, "fp16 wgrad")
    close(db, refb, 1e-3, 2e-4 * refb.abs().max().item(), "fp16 bias gradient")
    # with a loss scale installed the outputs leave multiplied by 1 / S (exactly: S is a power of two)
    st = ops.new_loss_scale_state(DEV)
    st[0], st[1] = 1024.0, 1.0 / 1024.0
    dW2, db2 = torch.empty_like(dW), torch.empty_like(db)
    with ops.loss_scale(st.data_ptr()):
        ops.wgrad(dY.to(DEV), X.to(DEV), dW2, M, ws, dbias=db2)
    assert torch.equal(dW2 * 1024.0, dW) and torch.equal(db2 * 1024.0, db)
    dW3 = torch.empty_like(dW)
    ops.wgrad(dY.to(DEV), X.to(DEV), dW3, M, ws)
    assert torch.equal(dW3, dW), "the scale must not outlive its context"


def test_wgrad_group_fp16_on_256_tiles_with_inverse_scale():
    """A layer's four weight gradients as one group in the fp16 format: every side a multiple of 256 -> 256 x 256 tiles (round 4), token splits with
    slabs (the reduction applies 1 / S), a token tail, bias gradients on two of them."""
    shapes = [(4100, 768, 768), (4100, 2304, 768), (4100, 3072, 768), (4100, 768, 3072)]
    st = ops.new_loss_scale_state(DEV)
    st[0], st[1] = 256.0, 1.0 / 256.0
    q, refs = ops.WgradQueue(), []
    for i, (M, N1, N2) in enumerate(shapes):
        dY, X = rnd(30 + i, (M, N1)).half(), rnd(40 + i, (M, N2)).half()
        dW = torch.full((N1, N2), float("nan"), device=DEV)
        db = torch.full((N1,), float("nan"), device=DEV) if i % 2 == 0 else None
        q.add(dY.to(DEV), X.to(DEV), dW, M, dbias=db)
        refs.append((dW, db, dY.double().T @ X.double() / 256.0, dY.double().sum(0) / 256.0))
    with ops.loss_scale(st.data_ptr()):
        q.flush(accumulate=False)
    for dW, db, rW, rb in refs:
        close(dW, rW, 1e-3, 2e-4 * rW.abs().max().item(), "fp16 wgrad group")
        if db is not None:
            close(db, rb, 1e-3, 2e-4 * rb.abs().max().item(), "fp16 wgrad group bias")


def _attn_ref(qkv, mask, nseq, L, H):
    d = H * 64
    x = qkv.double().view(nseq, L, 3, H, 64)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    s = q @ k.transpose(2, 3) * 0.125
    if mask is not None:
        s = s.masked_fill(mask[:, None, None, :] == 0, -1e30)
    p = torch.softmax(s, -1)
    return (p @ v).transpose(1, 2).reshape(nseq * L, d), torch.logsumexp(s, -1)


@pytest.mark.parametrize("nseq,L,H", [(3, 32, 2), (2, 100, 3), (48, 128, 12), (2, 256, 2), (1, 200, 1), (43, 96, 12)])
def test_attention_fwd_bwd_fp16(nseq, L, H):
    """fp16 q / k / v, context, dO and dq / dk / dv through every kernel of the family: one item per workgroup (few items), the persistent
    kernels (many items, L <= 128) and the streaming kernels (L > 128), against fp64 autograd on the same fp16 inputs."""
    d, T = H * 64, nseq * L
    qkv = rnd(10, (T, 3 * d)).half()
    lens = np.clip(syn.msmarco_lengths(11, nseq, L), 2, L)
    lens[0] = L
    mask = torch.from_numpy((np.arange(L)[None, :] < lens[:, None]).astype(np.int64))
    qv = qkv.double().requires_grad_(True)
    ref, ref_lse = _attn_ref(qv, mask, nseq, L, H)
    dctx = rnd(12, (T, d)).half()
    ref.backward(dctx.double())
    ctx = torch.empty(T, d, dtype=torch.float16, device=DEV)
    lse = torch.empty(nseq, H, L, dtype=torch.float32, device=DEV)
    ops.attention_fwd(qkv.to(DEV), mask.to(DEV), ctx, lse, nseq, L, H, full_family=True)
    close(ctx, ref, 1 / 512, 3e-3, f"fp16 attention fwd L={L}")
    close(lse, ref_lse, 1e-4, 1e-3, "lse")
    dqkv = torch.zeros(T, 3 * d, dtype=torch.float16, device=DEV)
    ops.attention_bwd(qkv.to(DEV), mask.to(DEV), ctx, dctx.to(DEV), lse, dqkv, nseq, L, H)
    g = qv.grad.float()
    close(dqkv, g, 1 / 256, 3e-3 * g.abs().max().item(), f"fp16 attention bwd L={L}")


@pytest.mark.parametrize("nseq,L,H,p", [(48, 128, 12, 0.1), (70, 64, 8, 0.1)])
def test_attention_fp16_dropout_bits_match_the_hash(nseq, L, H, p):
    """The fp16 persistent forward leaves the same keep bits as the bf16 one computes; the fp16 backward gives the same result from the bits
    as from re-hashing, and the persistent backward equals the one-item kernel bit for bit."""
    d, T = H * 64, nseq * L
    g = torch.Generator(device=DEV).manual_seed(7)
    qkv = torch.randn(T, 3 * d, device=DEV, generator=g).half()
    dctx = torch.randn(T, d, device=DEV, generator=g).half()
    mask = torch.ones(nseq, L, dtype=torch.int64, device=DEV)
    ctx = torch.empty(T, d, dtype=torch.float16, device=DEV)
    lse = torch.empty(nseq, H, L, dtype=torch.float32, device=DEV)
    bits = ops.attention_drop_bits(nseq, L, H, p, DEV)
    assert bits is not None
    ops.attention_fwd(qkv, mask, ctx, lse, nseq, L, H, dropout_p=p, seed=99, drop_bits=bits)
    outs = []
    for two_role, b in ((0, None), (1, None), (1, bits)):
        ops.set_tuning("attn_bwd2", two_role)
        dq = torch.full((T, 3 * d), float("nan"), dtype=torch.float16, device=DEV)
        try:
            ops.attention_bwd(qkv, mask, ctx, dctx, lse, dq, nseq, L, H, dropout_p=p, seed=99, drop_bits=b)
        finally:
            ops.set_tuning("attn_bwd2", 1)
        outs.append(dq)
    assert torch.isfinite(outs[0].float()).all()
    assert torch.equal(outs[0], outs[1])
    # bits vs hash: dQ and dK bit for bit.  dV differs in the LAST fp16 bit of <0.5 % of its elements: with the hash the compiler rounds
    # p * 1/(1-p_drop) to fp16 ONCE (v_fma_mix), with the bits the AND sits between the fp32 multiply and the conversion (two roundings), so
    # about one probability in 2^14 differs by an fp16 ulp (measured in round 4: 431 of 73728 (key, head) rows, max |diff| one ulp).
    # A wrong keep bit would move dV by p * dO ~ 1e-2, forty ulps.
    assert torch.equal(outs[1][:, :2 * d], outs[2][:, :2 * d])
    dv1, dv2 = outs[1][:, 2 * d:].float(), outs[2][:, 2 * d:].float()
    diff = (dv1 - dv2).abs()
    assert float((diff > 0).float().mean()) < 5e-3
    # one fp16 ulp of the result, or (where dV cancels to ~0) one ulp of a probability times |dO|
    assert bool((diff <= torch.maximum(dv1.abs(), dv2.abs()) * 2.0 ** -9 + 2.0 ** -11).all())     # (two ulps: a peaked row can hold two such probabilities)


@pytest.mark.parametrize("nseq,L,H", [(5, 40, 2), (3, 200, 3)])
def test_attention_cls_bwd_fp16(nseq, L, H):
    d = H * 64
    kv = rnd(20, (nseq * L, 2 * d)).half()
    qc = rnd(21, (nseq, d)).half()
    dctx = rnd(22, (nseq, d)).half()
    mask = torch.ones(nseq, L, dtype=torch.int64)
    k = kv.double().view(nseq, L, 2, H, 64)
    kk, vv = k[:, :, 0].requires_grad_(True), k[:, :, 1].requires_grad_(True)
    q = qc.double().view(nseq, H, 64).requires_grad_(True)
    s = torch.einsum("nhd,nlhd->nhl", q, kk) * 0.125
    pr = torch.softmax(s, -1)
    out = torch.einsum("nhl,nlhd->nhd", pr, vv).reshape(nseq, d)
    out.backward(dctx.double())
    ctx = torch.empty(nseq, d, dtype=torch.float16, device=DEV)
    probs = torch.empty(nseq, H, L, dtype=torch.float32, device=DEV)
    ops.attention_cls_fwd(qc.to(DEV), kv.to(DEV), mask.to(DEV), ctx, probs, nseq, L, H)
    close(ctx, out, 1 / 512, 3e-3, "fp16 CLS attention fwd")
    dqc = torch.empty(nseq, d, dtype=torch.float16, device=DEV)
    dkv = torch.empty(nseq * L, 2 * d, dtype=torch.float16, device=DEV)
    ops.attention_cls_bwd(qc.to(DEV), kv.to(DEV), probs, dctx.to(DEV), dqc, dkv, nseq, L, H)
    close(dqc, q.grad.reshape(nseq, d), 1 / 256, 3e-3 * q.grad.abs().max().item(), "dq")
    ref_dkv = torch.stack([kk.grad, vv.grad], dim=2).reshape(nseq * L, 2 * d)
    close(dkv, ref_dkv, 1 / 256, 3e-3 * ref_dkv.abs().max().item(), "dk | dv")


@pytest.mark.parametrize("M,N,K", [(2048, 768, 3072), (4100, 3072, 768), (240, 768, 768), (256, 3072, 768)])
def test_gemm_backward_flavours_in_fp16(M, N, K):
    """The data-gradient flavours on fp16 operands (ring kernel for M >= 1024, small-M kernel with and without split-K): plain, gelu'
    multiply from an fp16 tape tensor, fp32 out, fp32 residual in + fp32 out; and the forward flavour that writes gelu'(x) as fp16."""
    A, B = rnd(31, (M, K)).half(), rnd(32, (N, K), 0.05).half()
    base = A.double() @ B.double().T
    Ad, Bd = A.to(DEV), B.to(DEV)
    tol = dict(rtol=1 / 512, atol=2e-3 * base.abs().max().item())
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ops.gemm_nt(Ad, Bd, out)
    close(out, base, what="plain", **tol)
    gp = (torch.rand(M, N) * 1.2 - 0.1).half()
    ops.gemm_nt(Ad, Bd, out, gelu_pre=gp.to(DEV), act=2)
    close(out, base * gp.double(), what="gelu' multiply (fp16 tape)", **tol)
    o32 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm_nt(Ad, Bd, o32)
    close(o32, base, 1e-5, 2e-4 * math.sqrt(K) * 0.05, "fp32 out")
    res = rnd(33, (M, N))
    ops.gemm_nt(Ad, Bd, o32, residual=res.to(DEV))
    close(o32, base + res.double(), 1e-5, 2e-4 * math.sqrt(K) * 0.05, "fp32 residual + fp32 out")
    bias = rnd(34, (N,))
    pre = torch.full((M, N), float("nan"), dtype=torch.float16, device=DEV)
    ops.gemm_nt(Ad, Bd, out, bias=bias.to(DEV), preact=pre, act=3)
    x = base + bias.double()
    close(out, torch.nn.functional.gelu(x), what="gelu forward", **tol)
    dref = 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)
    close(pre, dref, 1 / 512, 2e-3, "gelu'(x) on the fp16 tape")


@pytest.mark.parametrize("T,d,p", [(300, 768, 0.0), (1000, 768, 0.1), (130, 128, 0.1)])
def test_layernorm_bwd_fp16_operand_copy(T, d, p):
    """fp32 stream in / out as before; the MFMA operand copy and the branch term are fp16 (same values as the bf16 build up to their rounding);
    parameter-gradient sums leave multiplied by 1 / S."""
    import oracle.dropout_ref as DRo
    x, dy, br = rnd(41, (T, d)), rnd(42, (T, d)), rnd(43, (T, d), 0.3).half()
    gamma = 1 + 0.1 * rnd(44, (d,))
    mean, var = x.mean(1), x.var(1, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-12)
    outs = {}
    for fmt in (torch.float16, torch.bfloat16):
        dx, d2 = torch.empty(T, d, device=DEV), torch.empty(T, d, dtype=fmt, device=DEV)
        dg, db, dbi = torch.empty(d, device=DEV), torch.empty(d, device=DEV), torch.empty(d, device=DEV)
        part = torch.empty(ops.ln_partial_elems(T, d), device=DEV)
        ops.layernorm_bwd(dy.to(DEV), x.to(DEV), mean.to(DEV), rstd.to(DEV), gamma.to(DEV), dx, d2, dg, db, dbi, part, T, dropout_p=p, seed=5,
                          accumulate=False, dy_branch=br.to(DEV).to(fmt))
        outs[fmt] = (dx, d2, dg, db, dbi)
    h, b = outs[torch.float16], outs[torch.bfloat16]
    g = dy.double() + br.double()
    xh = (x.double() - mean.double()[:, None]) * rstd.double()[:, None]
    t = g * gamma.double()
    ref_dx = rstd.double()[:, None] * (t - t.mean(1, keepdim=True) - xh * (t * xh).mean(1, keepdim=True))
    close(h[0], ref_dx, 1e-4, 1e-4, "dx (fp32 stream)")
    keep = torch.from_numpy(DRo.keep_mask(5, p, T, d)) if p > 0 else torch.ones(T, d, dtype=torch.bool)
    close(h[1], torch.where(keep, ref_dx / (1 - p), torch.zeros(())), 1 / 1024, 1e-4, "fp16 operand copy")
    close(h[2], (g * xh).sum(0), 1e-4, 1e-3, "dgamma")
    close(h[3], g.sum(0), 1e-4, 1e-3, "dbeta")
    st = ops.new_loss_scale_state(DEV)
    st[0], st[1] = 256.0, 1.0 / 256.0
    dx, d2 = torch.empty(T, d, device=DEV), torch.empty(T, d, dtype=torch.float16, device=DEV)
    dg, db, dbi = torch.empty(d, device=DEV), torch.empty(d, device=DEV), torch.empty(d, device=DEV)
    with ops.loss_scale(st.data_ptr()):
        ops.layernorm_bwd(dy.to(DEV), x.to(DEV), mean.to(DEV), rstd.to(DEV), gamma.to(DEV), dx, d2, dg, db, dbi,
                          torch.empty(ops.ln_partial_elems(T, d), device=DEV), T, dropout_p=p, seed=5, accumulate=False, dy_branch=br.to(DEV))
    assert torch.equal(dx, h[0]) and torch.equal(d2, h[1]), "activation gradients keep the scale they came with"
    assert torch.allclose(dg * 256.0, h[2], rtol=1e-6, atol=0) and torch.allclose(db * 256.0, h[3], rtol=1e-6, atol=0)


def test_loss_scale_adapt_and_safety_net():
    """S = 2^(12 + h - ceil(log2 max|x|)): the largest element lands in [2^11, 2^12]; zeros and non-finite values give S = 1; a non-finite gradient
    norm skips the optimizer step and lowers the headroom exponent by 2, `interval` finite steps give 1 back."""
    st = ops.new_loss_scale_state(DEV)
    for amax in (3.7e-3, 1.0, 12.25, 900.0):
        a = torch.randn(8, 768, device=DEV)
        a *= amax / a.abs().max()
        b = torch.randn(256, 768, device=DEV) * 1e-3 * amax
        a0, b0 = a.clone(), b.clone()
        ops.loss_scale_adapt(a, b, st)
        S = st[0].item()
        assert S == 2.0 ** round(math.log2(S)) and st[1].item() == 1.0 / S
        assert 2048.0 <= a.abs().max().item() <= 4096.0, (amax, S, a.abs().max().item())
        assert torch.equal(a, a0 * S) and torch.equal(b, b0 * S)
    tiny = torch.full((4, 8), 3.7e-5, device=DEV)      # the exponent is clamped at 24: tiny gradients are not blown up without bound
    ops.loss_scale_adapt(tiny, None, st)
    assert st[0].item() == 2.0 ** 24
    z = torch.zeros(4, 8, device=DEV)
    ops.loss_scale_adapt(z, None, st)
    assert st[0].item() == 1.0
    # the safety net: clip_coef under the scale context
    clip = torch.zeros(3, device=DEV)
    partial = torch.empty(ops.sqnorm_blocks(), device=DEV)
    g = torch.ones(1024, device=DEV)
    with ops.loss_scale(st.data_ptr(), 3):
        g[5] = float("inf")
        ops.grad_clip_coef(g, 1.0, partial, clip)
        assert clip[2].item() == 1.0 and st[4].item() == -2.0 and st[3].item() == 1.0
        g[5] = 1.0
        for i in range(3):
            ops.grad_clip_coef(g, 1.0, partial, clip)
        assert clip[2].item() == 0.0 and st[4].item() == -1.0 and st[2].item() == 0.0
    a = torch.full((4, 8), 1.0, device=DEV)
    ops.loss_scale_adapt(a, None, st)
    assert st[0].item() == 2.0 ** (12 - 1 - 1)          # frexp(1.0) = 0.5 * 2^1: e = 1; h = -1


def test_non_finite_gradients_skip_the_step_and_training_continues(monkeypatch):
    """An overflow in the fp16 backward must cost one step, not the run: parameters, moments untouched, the next step proceeds with more headroom."""
    monkeypatch.setenv("CLDRD_GRAPH", "0")
    cfg = EncoderConfig(arch="distilbert", vocab_size=512, dim=128, n_heads=2, hidden_dim=256, n_layers=2, max_position_embeddings=64,
                        dropout=0.0, attention_dropout=0.0)
    model = selftest.build_tiny_model(cfg).cuda().train()
    tr = NwayTrainer(model, loss="kl_div", learning_rate=1e-3, warmup_steps=0, total_steps=20)
    assert tr.amp16
    batch = syn.nway_batch(4680, 3, 4, 8, 16, vocab=cfg.vocab_size, ragged=True, label_kind="teacher")
    tr.train_step(batch)
    p1, m1 = tr.flat_p.clone(), tr.m.clone()
    real = tr._norm_launches

    def poisoned():
        # a non-finite gradient shows up as a non-finite partial sum of squares - from a separate norm pass over flat_g, or (round 5) from the
        # kernels that wrote the gradients (norm sink): poison both, whichever path this configuration takes
        tr.flat_g[-1] = float("nan")
        tr.norm_partial[ops.sqnorm_blocks() // 2] = float("nan")
        real()
    tr._norm_launches = poisoned
    tr.use_norm_sink = False      # (test hook) the late piece by a separate pass: it then sees the poisoned flat_g
    tr.train_step(batch)
    torch.cuda.synchronize()
    assert tr.clip[2].item() == 1.0 and torch.equal(tr.flat_p, p1) and torch.equal(tr.m, m1)
    assert tr._scale_state[3].item() == 1.0 and tr._scale_state[4].item() == -2.0
    tr._norm_launches = real
    tr.use_norm_sink = True
    tr.train_step(batch)
    torch.cuda.synchronize()
    assert tr.clip[2].item() == 0.0 and not torch.equal(tr.flat_p, p1) and torch.isfinite(tr.flat_p).all()


@pytest.mark.parametrize("arch", ["distilbert", "bert"])
def test_amp16_and_bf16_base_modes_agree_to_rounding(arch, monkeypatch):
    """The two training modes compute the same function: logits within fp16 / bf16 operand rounding of each other, gradients with cosine
    >= 0.9995 per tower - and they are different code paths (not bit-identical)."""
    cfg = EncoderConfig(arch=arch, vocab_size=512, dim=128, n_heads=2, hidden_dim=256, n_layers=3, max_position_embeddings=64,
                        dropout=0.0, attention_dropout=0.0)
    batch = syn.nway_batch(4680, 3, 4, 10, 32, vocab=cfg.vocab_size, ragged=True)
    res = {}
    for amp in ("fp16", "bf16"):
        monkeypatch.setenv("CLDRD_AMP", amp)
        model = selftest.build_tiny_model(cfg).cuda().train()
        tr = NwayTrainer(model, loss="margin_mse")
        assert tr.amp16 == (amp == "fp16")
        _, lg = tr.forward_backward(batch)
        torch.cuda.synchronize()
        res[amp] = (lg.clone(), tr.flat_g.double().clone())
    (l16, g16), (lb, gb) = res["fp16"], res["bf16"]
    assert not torch.equal(l16, lb)
    assert (l16 - lb).abs().max().item() <= 2e-2 * lb.abs().max().item()
    c = torch.nn.functional.cosine_similarity(g16, gb, dim=0).item()
    assert c >= 0.9995 and abs((g16.norm() / gb.norm()).item() - 1.0) <= 5e-3, c


def test_evaluation_after_fp16_steps_sees_current_bf16_weights(monkeypatch):
    """AdamW does not write the bf16 weight shadow in the all-fp16 mode (no pass of a training step reads it); an evaluation forward, which
    does, must cast it first: embeddings after two steps equal those from explicitly refreshed shadows, and differ from the pre-step ones."""
    monkeypatch.setenv("CLDRD_GRAPH", "0")
    cfg = EncoderConfig(arch="distilbert", vocab_size=512, dim=128, n_heads=2, hidden_dim=256, n_layers=2, max_position_embeddings=64,
                        dropout=0.0, attention_dropout=0.0)
    model = selftest.build_tiny_model(cfg).cuda().train()
    tr = NwayTrainer(model, loss="kl_div", learning_rate=1e-3, warmup_steps=0, total_steps=20)
    assert tr.amp16
    batch = syn.nway_batch(4680, 3, 4, 8, 16, vocab=cfg.vocab_size, ragged=True, label_kind="teacher")
    enc = {k: v.view(-1, v.shape[-1]).cuda() for k, v in batch["nway_passages"].items()}
    model.eval()
    with torch.no_grad():
        e0 = model.passage_embs(enc).clone()
    model.train()
    for _ in range(2):
        tr.train_step(batch)
    assert all(getattr(t, "_h_stale", False) for t in model.towers())
    model.eval()
    with torch.no_grad():
        e1 = model.passage_embs(enc).clone()
        assert not any(getattr(t, "_h_stale", False) for t in [model.passage_encoder])
        for t in model.towers():
            t.refresh_shadows(need_transposed=True)
        e2 = model.passage_embs(enc).clone()
    assert torch.equal(e1, e2) and not torch.equal(e0, e1)
    model.train()
    tr.train_step(batch)            # and training goes on from the refreshed state
    torch.cuda.synchronize()
    assert torch.isfinite(tr.flat_p).all()


def test_evaluation_between_replayed_steps_sees_current_weights():
    """train (graph replay), evaluate, train (replay), evaluate.  The captured AdamW skips the bf16 weight shadow; the Python
    bookkeeping that marks it stale ran at capture time only, so every replay has to mark the towers again - otherwise the SECOND evaluation
    encodes with the bf16 matrices cast at the first one.  Graph replay ON (the default)."""
    cfg = EncoderConfig(arch="distilbert", vocab_size=512, dim=128, n_heads=2, hidden_dim=256, n_layers=2, max_position_embeddings=64,
                        dropout=0.0, attention_dropout=0.0)
    model = selftest.build_tiny_model(cfg).cuda().train()
    tr = NwayTrainer(model, loss="kl_div", learning_rate=1e-3, warmup_steps=0, total_steps=50)
    assert tr.amp16
    batch = syn.nway_batch(4680, 3, 4, 8, 16, vocab=cfg.vocab_size, ragged=False, label_kind="teacher")
    batch = {k: ({kk: vv.cuda() for kk, vv in v.items()} if isinstance(v, dict) else v.cuda()) for k, v in batch.items()}
    enc = {k: v.view(-1, v.shape[-1]) for k, v in batch["nway_passages"].items()}

    def evaluate():
        model.eval()
        with torch.no_grad():
            e = model.passage_embs(enc).clone()
        model.train()
        return e
    for _ in range(6):                     # three eager steps, the capture, replays
        tr.train_step(batch)
    assert any(e["graph"] is not None for e in tr._graphs.values()), "the step was not captured"
    e1 = evaluate()
    for _ in range(3):
        tr.train_step(batch)               # replays only
    assert all(getattr(t, "_h_stale", False) for t in model.towers()), "a replay left the bf16 shadow marked fresh"
    e2 = evaluate()
    for t in model.towers():
        t.refresh_shadows(need_transposed=True)
    e3 = evaluate()
    assert torch.equal(e2, e3), "the evaluation after replayed steps used stale bf16 weights"
    assert not torch.equal(e1, e2)


@pytest.mark.parametrize("graph", ["0", "1"])
def test_skipped_steps_do_not_advance_adams_bias_correction(graph, monkeypatch):
    """GradScaler semantics (reference nway_listwise_1.py:357): scaler.step() does not call optimizer.step() on a non-finite gradient, so the
    per-parameter Adam `step` - the bias-correction exponent - counts applied steps only.  Two trainers from the same state: A runs steps
    1..3, B runs a poisoned (skipped) step followed by the same three batches; B's parameters after its 4 calls must equal A's after 3 (the
    lr schedule is flat here), and the checkpointed optimizer step is 3 in both."""
    monkeypatch.setenv("CLDRD_GRAPH", graph)
    cfg = EncoderConfig(arch="distilbert", vocab_size=512, dim=128, n_heads=2, hidden_dim=256, n_layers=2, max_position_embeddings=64,
                        dropout=0.0, attention_dropout=0.0)
    batch = syn.nway_batch(4680, 3, 4, 8, 16, vocab=cfg.vocab_size, ragged=False, label_kind="teacher")
    batch = {k: ({kk: vv.cuda() for kk, vv in v.items()} if isinstance(v, dict) else v.cuda()) for k, v in batch.items()}

    def make():
        model = selftest.build_tiny_model(cfg).cuda().train()
        # a constant lr (total_steps huge, no warm-up): the only thing that can differ between A and B is the bias-correction exponent
        return NwayTrainer(model, loss="kl_div", learning_rate=1e-3, warmup_steps=0, total_steps=10 ** 9, weight_decay=0.0)
    A, Bt = make(), make()
    assert torch.equal(A.flat_p, Bt.flat_p)
    for _ in range(3):
        A.train_step(batch)
    real = Bt._norm_launches

    def poisoned():
        Bt.flat_g[-1] = float("nan")
        real()
    Bt._norm_launches = poisoned
    Bt.use_norm_sink = False      # (test hook) the poisoned step takes the separate norm pass (which re-reads flat_g)
    p0 = Bt.flat_p.clone()
    Bt.train_step(batch)                    # skipped
    torch.cuda.synchronize()
    assert torch.equal(Bt.flat_p, p0) and Bt.skipped_steps() == 1
    Bt._norm_launches = real
    Bt.use_norm_sink = True
    for _ in range(3):
        Bt.train_step(batch)
    torch.cuda.synchronize()
    # Not bit for bit: the embedding-table gradients are float atomics, B's loss scale is 4x smaller after its skipped step (other fp16
    # roundings), and Adam's first steps turn a last-bit difference of a near-zero gradient into +-lr on that element.  The UPDATE as a
    # whole must agree: with the host's step count as the exponent (2, 3, 4 instead of 1, 2, 3) B's three updates would be 0.74, 0.86 and
    # 0.91 of A's - 16 % short in norm.
    ua, ub = (A.flat_p - p0).double(), (Bt.flat_p - p0).double()
    rel = ((ua - ub).norm() / ua.norm()).item()
    short = 1.0 - (ub.norm() / ua.norm()).item()
    assert ua.norm().item() > 0 and rel <= 0.05 and abs(short) <= 0.02, (rel, short)
    sa, sb = A.optimizer_state_dict(), Bt.optimizer_state_dict()
    assert {v["step"] for v in sa["state"].values()} == {3} and {v["step"] for v in sb["state"].values()} == {3}
    # without the correction B's exponent would have been 2, 3, 4: its first applied update alone would differ by ~(1 - b1^2)/(1 - b1) ~ 1.9x


@pytest.mark.parametrize("T,d,p,branch", [(300, 768, 0.0, True), (1000, 768, 0.1, True), (130, 128, 0.1, False), (2050, 768, 0.0, False)])
def test_layernorm_bwd_fp16_gradient_stream(T, d, p, branch):
    """Round 5: the gradient stream itself is fp16 (dy in, dx out; the default of the fp16 mode).  Same arithmetic as the
    fp32-stream kernel on the same (fp16-representable) inputs: dx equals the fp32-stream dx rounded once to fp16, the dropped operand copy and the
    parameter-gradient sums are bit-identical; without dropout no second tensor is written at all."""
    x = rnd(51, (T, d))
    dy16 = rnd(52, (T, d)).half()
    br = rnd(53, (T, d), 0.3).half() if branch else None
    gamma = 1 + 0.1 * rnd(54, (d,))
    mean, var = x.mean(1), x.var(1, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-12)
    args = (x.to(DEV), mean.to(DEV), rstd.to(DEV), gamma.to(DEV))
    brd = br.to(DEV) if branch else None

    def run(dy, stream16, with_copy):
        dx = torch.full((T, d), float("nan"), dtype=torch.float16 if stream16 else torch.float32, device=DEV)
        d2 = torch.full((T, d), float("nan"), dtype=torch.float16, device=DEV) if with_copy else None
        dg, db, dbi = torch.empty(d, device=DEV), torch.empty(d, device=DEV), torch.empty(d, device=DEV)
        ops.layernorm_bwd(dy, *args, dx, d2, dg, db, dbi, torch.empty(ops.ln_partial_elems(T, d), device=DEV), T, dropout_p=p, seed=9,
                          accumulate=False, dy_branch=brd)
        return dx, d2, dg, db, dbi
    ref = run(dy16.float().to(DEV), False, True)                 # the fp32-stream kernel on the same values
    got = run(dy16.to(DEV), True, True)
    assert torch.equal(got[0], ref[0].half()), "fp16 stream dx is not the fp32-stream dx rounded once"
    assert torch.equal(got[1], ref[1]), "operand copy differs"
    for i in (2, 3, 4):
        assert torch.equal(got[i], ref[i]), "parameter-gradient sums differ"
    if p == 0.0:
        only = run(dy16.to(DEV), True, False)                   # no dropout: the stream tensor is the operand
        assert only[1] is None and torch.equal(only[0], got[0]) and torch.equal(only[0], got[1])
        assert torch.equal(only[4], got[4])
    with pytest.raises((TypeError, ValueError)):
        ops.layernorm_bwd(dy16.to(DEV), *args, torch.empty(T, d, device=DEV), None, None, None, None,
                          torch.empty(ops.ln_partial_elems(T, d), device=DEV), T)          # fp16 dy with an fp32 dx


def test_fp16_stream_row_helpers():
    """scatter_cls_grad(_idx) into an fp16 stream tensor; fp16 dst += fp32 src for the CLS rows (one rounding)."""
    R, d, L = 6, 128, 5
    dcls = rnd(61, (R, d)).to(DEV)
    g = torch.full((R * L, d), float("nan"), dtype=torch.float16, device=DEV)
    ops.scatter_cls_grad(dcls, g, R, L, R * L)
    want = torch.zeros(R * L, d, dtype=torch.float16, device=DEV)
    want[::L] = dcls.half()
    assert torch.equal(g, want)
    idx = torch.tensor([0, 3, 4, 9, 17, 29], dtype=torch.int32, device=DEV)
    g2 = torch.full((R * L, d), float("nan"), dtype=torch.float16, device=DEV)
    ops.scatter_cls_grad_idx(dcls, g2, idx, R * L)
    want2 = torch.zeros(R * L, d, dtype=torch.float16, device=DEV)
    want2[idx.long()] = dcls.half()
    assert torch.equal(g2, want2)
    base = rnd(62, (R * L, d)).half().to(DEV)
    src = rnd(63, (R, d)).to(DEV)
    a = base.clone()
    ops.add_rows_strided(a, src, R, L)
    w = base.clone()
    w[::L] = (base[::L].float() + src).half()
    assert torch.equal(a, w)
    b = base.clone()
    ops.add_rows_idx(b, src, idx, R)
    w2 = base.clone()
    w2[idx.long()] = (base[idx.long()].float() + src).half()
    assert torch.equal(b, w2)


def test_fp16_and_fp32_gradient_streams_agree_end_to_end(monkeypatch):
    """The whole backward of a small model with the fp16 stream (default) and with the fp32 stream (test hook `_grad_stream16 = False`) from the same state: every
    parameter-gradient tensor agrees to fp16 rounding noise (cosine >= 0.99999 per tensor, norms to 1e-3)."""
    cfg = EncoderConfig(arch="distilbert", vocab_size=512, dim=128, n_heads=2, hidden_dim=256, n_layers=3, max_position_embeddings=64,
                        dropout=0.1, attention_dropout=0.1)
    batch = syn.nway_batch(4680, 4, 6, 10, 40, vocab=cfg.vocab_size, ragged=True, label_kind="teacher")
    grads = {}
    for mode in ("fp16", "fp32"):
        monkeypatch.setenv("CLDRD_GRAPH", "0")
        model = selftest.build_tiny_model(cfg, seed=5).cuda().train()
        for t in model.towers():
            t._grad_stream16 = mode == "fp16"
        tr = NwayTrainer(model, loss="kl_div", learning_rate=1e-4, warmup_steps=0, total_steps=10)
        assert tr.amp16 and all(t.grad_stream16 == (mode == "fp16") for t in model.towers())
        for t in model.towers():
            t.step_seed = 11
        tr.forward_backward(batch)
        torch.cuda.synchronize()
        grads[mode] = {n: p.grad.detach().double().clone() for n, p in model.named_parameters()}
    worst = 1.0
    for n, a in grads["fp16"].items():
        b = grads["fp32"][n]
        if b.norm().item() < 1e-9:
            continue
        c = float((a * b).sum() / (a.norm() * b.norm()))
        worst = min(worst, c)
        if "k_lin.bias" in n:           # mathematically zero gradient: noise in every implementation
            continue
        assert c >= 0.99999, (n, c)
        assert abs(a.norm().item() / b.norm().item() - 1.0) <= 1e-3, n
    print("worst cosine fp16-stream vs fp32-stream:", worst)


@pytest.mark.parametrize("graph", ["0", "1"])
def test_clip_norm_from_the_kernels_that_write_the_gradients(graph, monkeypatch):
    """Round 5 (norm sink): the passage tower's layer gradients are complete only behind its last weight-gradient group; their share of the clip
    norm now comes from that group's slab reduction and from the LayerNorm-parameter reduction (sums of squares of what they write) instead of
    a separate pass.  DistilBERT-width model at a token count where the group splits the tokens (slabs): the sink must be in use, the norm
    must equal the norm of the whole gradient buffer, and the step must equal the step taken with the separate pass (`use_norm_sink = False`)
    outside the embedding atomics."""
    monkeypatch.setenv("CLDRD_GRAPH", graph)
    cfg = EncoderConfig(arch="distilbert", vocab_size=2048, dim=768, n_heads=12, hidden_dim=3072, n_layers=2, max_position_embeddings=128,
                        dropout=0.1, attention_dropout=0.1)
    batch = syn.nway_batch(4680, 8, 32, 16, 128, vocab=cfg.vocab_size, ragged=False, label_kind="teacher")
    batch = {k: ({kk: vv.cuda() for kk, vv in v.items()} if isinstance(v, dict) else v.cuda()) for k, v in batch.items()}
    res = {}
    for sink in ("1", "0"):
        model = selftest.build_tiny_model(cfg, seed=2).cuda().train()
        tr = NwayTrainer(model, loss="kl_div", learning_rate=1e-4, warmup_steps=0, total_steps=100)
        tr.use_norm_sink = sink == "1"
        for _ in range(5):
            tr.train_step(batch)
        torch.cuda.synchronize()
        used = int(getattr(model.passage_encoder, "norm_sink_used", -1))
        if sink == "1":
            assert tr._sink_on and used > 0, "the weight-gradient group did not take the slab path at this size: the sink was not exercised"
        norm_buf = tr.flat_g.double().norm().item()
        assert abs(tr.clip[0].item() - norm_buf) <= 2e-6 * norm_buf, (sink, tr.clip[0].item(), norm_buf)
        res[sink] = (tr.clip.clone(), tr.flat_p.clone())
    # (each mode's norm is checked against its OWN gradient buffer above, to 2e-6: that is the comparison that pins the sink.  The two
    # trainers' trajectories differ after five steps - float atomics in the embedding gradients - so across them only the scale is compared)
    a, b = res["1"], res["0"]
    assert abs(a[0][1].item() - b[0][1].item()) <= 2e-2 * b[0][1].item()
    assert (a[1] - b[1]).double().norm().item() / b[1].double().norm().item() <= 1e-3
