#!/usr/bin/env python3
"""Randomised check of the loss kernels against the oracle (oracle/losses_ref.py, itself pinned to the reference's losses by goldens):
random B x N (1 .. 24 x 1 .. 400), score scales, label patterns (teacher scores, graded relevance, binary, ties, all-equal rows,
-1 padding at random places / whole rows), every loss kind, reduction and lambda_loss scheme.  usage: tools/loss_fuzz.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cldrd_amd import hip_ops as ops
from oracle import losses_ref as LR
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
DEV = "cuda"
bad = 0
SCHEMES = [None, "ndcgLoss1_scheme", "ndcgLoss2_scheme", "lambdaRank_scheme", "ndcgLoss2PP_scheme", "rankNet_scheme",
           "rankNetWeightedByGTDiff_scheme", "rankNetWeightedByGTDiffPowed_scheme"]
def check(tag, val, grad, rv, rg, rtol=2e-4):
    global bad
    val, grad = float(val), grad.cpu().numpy().astype(np.float64)
    ok_v = (np.isnan(val) and np.isnan(rv)) or abs(val - rv) <= rtol * max(abs(rv), 1e-6) + 1e-6
    scale = max(np.abs(rg).max(), 1e-12) if np.isfinite(rg).all() else 1.0
    ok_g = np.array_equal(np.isnan(grad), np.isnan(rg)) and np.allclose(np.nan_to_num(grad), np.nan_to_num(rg), rtol=rtol, atol=rtol * scale + 3e-8)
    if not (ok_v and ok_g):
        bad += 1
        if bad < 15:
            print(f"MISMATCH {tag}: loss {val} vs {rv}; max grad diff {np.nanmax(np.abs(grad - rg)):.3e} (scale {scale:.3e})", flush=True)
for c in range(cases):
    B, N = int(rng.integers(1, 25)), int(rng.choice([1, 2, 3, 8, 32, 33, 100, 200, 400]))
    yp = (rng.standard_normal((B, N)) * float(rng.choice([0.1, 1.0, 10.0, 40.0]))).astype(np.float32)
    lab = str(rng.choice(["teacher", "graded", "binary", "ties", "equal"]))
    if lab == "teacher": yt = (rng.standard_normal((B, N)) * 5 + 10).astype(np.float32)
    elif lab == "graded": yt = rng.integers(0, 4, (B, N)).astype(np.float32)
    elif lab == "binary": yt = (rng.random((B, N)) < 0.1).astype(np.float32)
    elif lab == "ties": yt = np.round(rng.standard_normal((B, N))).astype(np.float32) + 2
    else: yt = np.full((B, N), 1.0, np.float32)
    padk = str(rng.choice(["none", "tail", "random", "row"]))
    pad_ok = lab != "teacher"
    if pad_ok and padk == "tail":
        for b in range(B): yt[b, int(rng.integers(1, N + 1)):] = -1
    elif pad_ok and padk == "random": yt[rng.random((B, N)) < 0.2] = -1
    elif pad_ok and padk == "row": yt[int(rng.integers(0, B))] = -1
    if lab == "ties": yp[:, : N // 2] = np.round(yp[:, : N // 2])             # tied predictions too
    red = str(rng.choice(["mean", "sum"]))
    P, Tt = torch.from_numpy(yp).to(DEV), torch.from_numpy(yt).to(DEV)
    tag = f"case {c} B {B} N {N} {lab} pad {padk} {red}"
    T = float(rng.choice([1.0, 0.5, 2.0]))
    if (yt >= 0).all():
        out, g = ops.loss_fwd_bwd("kl_div", P, Tt, T=T); check(tag + " kl", out[0], g, *LR.kl_div(yp, yt, T))
        out, g = ops.loss_fwd_bwd("margin_mse", P, Tt); check(tag + " mse", out[0], g, *LR.margin_mse(yp, yt))
        out, g = ops.loss_fwd_bwd("weighted_pointwise", P, Tt, T=T); check(tag + " wp", out[0], g, *LR.weighted_pointwise(yp, yt, T))
    if not (yt == -1).any():        # the reference's ranknet asserts that no label is the padding value (losses/ranknet.py:16)
        out, g = ops.loss_fwd_bwd("ranknet", P, Tt, reduction=red); check(tag + " ranknet", out[0], g, *LR.ranknet(yp, yt, reduction=red))
    out, g = ops.loss_fwd_bwd("lambda_mrr", P, Tt, reduction=red); check(tag + " lambda_mrr", out[0], g, *LR.lambda_mrr(yp, yt, reduction=red))
    if not (yt == -1).any():        # bweight_lambda_mrr asserts the same (losses/lambda_rank.py:18)
        bw = rng.random(B).astype(np.float32) + 0.5
        out, g = ops.loss_fwd_bwd("lambda_mrr", P, Tt, batch_weight=torch.from_numpy(bw).to(DEV), reduction=red)
        check(tag + " bweight", out[0], g, *LR.bweight_lambda_mrr(yp, yt, bw, reduction=red))
    sch = SCHEMES[int(rng.integers(0, len(SCHEMES)))]
    k = None if rng.random() < 0.5 else int(rng.integers(1, N + 1))
    kw = dict(weighing_scheme=sch, k=k, sigma=float(rng.choice([1.0, 0.5])), mu=10.0, reduction=red, reduction_log=str(rng.choice(["natural", "binary"])))
    try:
        rv, rg = LR.lambda_loss(yp, yt, **kw)
    except Exception as e:
        rv = None
    if rv is not None:
        out, g = ops.lambda_loss_fwd_bwd(P, Tt, **kw); check(tag + f" lambda_loss {sch} k={k}", out[0], g, rv, rg, rtol=1e-3)
print(f"{cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
