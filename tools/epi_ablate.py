#!/usr/bin/env python3
"""What the fused epilogues of the ring GEMM cost, per flavour of the all-fp16 training step (development build, CLDRD_GEMM_ABLATE_EPI:
7 = no epilogue at all, 8 = the whole epilogue but every tile reads / writes the FIRST tile's place, i.e. no HBM streams; wrong results).
T = 32768 rows, fp16 operands.  One process per mode (the mode is read per launch, the process split keeps clocks comparable)."""
import os, subprocess, sys
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _devlib  # noqa: F401
    import torch
    from cldrd_amd import hip_ops as ops
    T, d, f = 32768, 768, 3072
    dev = "cuda"
    h16 = lambda *s: (torch.randn(*s, device=dev) * 0.5).half()
    x, hbig = h16(T, d), h16(T, f)
    W1, W2, Wo = h16(f, d) * 0.05, h16(d, f) * 0.05, h16(d, d) * 0.05
    b_f, b_d = torch.randn(f, device=dev), torch.randn(d, device=dev)
    s32 = torch.randn(T, d, device=dev)
    mean, rstd = torch.zeros(T, device=dev), torch.ones(T, device=dev)
    gam, bet = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    out_f, pre = torch.empty(T, f, device=dev, dtype=torch.float16), torch.empty(T, f, device=dev, dtype=torch.float16)
    out32, out_d = torch.empty(T, d, device=dev), torch.empty(T, d, device=dev, dtype=torch.float16)
    cases = {
        "FFN1 fwd (bias, GELU, gelu' tape)  N3072 K768 ": (lambda: ops.gemm_nt(x, W1, out_f, T, bias=b_f, preact=pre, act=3), 2.0 * T * f * d),
        "FFN2 fwd (bias, drop, LN(res32), f32) N768 K3072": (lambda: ops.gemm_nt(hbig, W2, out32, T, bias=b_d, residual=s32, dropout_p=0.1, seed=5, residual_ln=(mean, rstd, gam, bet)), 2.0 * T * f * d),
        "out-proj (bias, drop, LN(res32), f32) N768 K768 ": (lambda: ops.gemm_nt(x, Wo, out32, T, bias=b_d, residual=s32, dropout_p=0.1, seed=5, residual_ln=(mean, rstd, gam, bet)), 2.0 * T * d * d),
        "dgrad FFN2 (x gelu' tape)           N3072 K768 ": (lambda: ops.gemm_nt(x, W1, out_f, T, gelu_pre=pre, act=2), 2.0 * T * f * d),
        "dgrad FFN1 (plain fp16 out)         N768 K3072": (lambda: ops.gemm_nt(hbig, W2, out_d, T), 2.0 * T * f * d),
    }
    res = []
    for name, (fn, fl) in cases.items():
        for _ in range(3): fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        res.append(f"{name}: {best * 1e3:6.1f} us")
    print(f"mode {sys.argv[1]}:\n   " + "\n   ".join(res))
else:
    names = {0: "full kernel", 7: "no epilogue", 8: "epilogue, every tile at the first tile's place (same-line conflicts)", 9: "epilogue, M panels folded onto the first four (cache-resident, no HBM streams)", 11: "epilogue, every tile at the place of tile (index mod 256): 32 MB footprint, no line shared between CUs"}
    for mode in (0, 7, 9, 11):
        env = dict(os.environ, CLDRD_GEMM_ABLATE_EPI=str(mode))
        r = subprocess.run([sys.executable, __file__, str(mode)], env=env, capture_output=True, text=True)
        print((r.stdout.strip() if r.stdout.strip() else r.stderr[-400:]), "  <-", names[mode], flush=True)
