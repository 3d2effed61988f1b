#!/usr/bin/env python3
"""In-kernel time stamps of the ring GEMM (development build, CLDRD_GEMM_ABLATE_EPI=10): where a tile's time goes, phase by phase, for the
fused flavours of the training step.  Wave 0 of the first 1024 workgroups stamps s_memtime (100 MHz) at kernel entry, after the first K tile
landed, after the K loop, after the epilogue barrier and after each of the four 32-row chunks of its epilogue."""
import ctypes, os, sys
os.environ["CLDRD_GEMM_ABLATE_EPI"] = "10"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401
import numpy as np
import torch
from cldrd_amd import hip_ops as ops, _lib
lib = _lib.load()
lib.cldrd_dev_stamps.restype = ctypes.POINTER(ctypes.c_ulonglong)
T, d, f = int(os.environ.get('T', 32768)), 768, 3072
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
    "FFN1 fwd (bias, GELU, gelu' tape) N3072 K768": lambda: ops.gemm_nt(x, W1, out_f, T, bias=b_f, preact=pre, act=3),
    "FFN1 eval (bias, GELU, no tape) N3072 K768": lambda: ops.gemm_nt(x, W1, out_f, T, bias=b_f, act=1),
    "FFN2 fwd (bias, drop, LN(res32), f32) N768 K3072": lambda: ops.gemm_nt(hbig, W2, out32, T, bias=b_d, residual=s32, dropout_p=0.1, seed=5, residual_ln=(mean, rstd, gam, bet)),
    "out-proj (bias, drop, LN(res32), f32) N768 K768": lambda: ops.gemm_nt(x, Wo, out32, T, bias=b_d, residual=s32, dropout_p=0.1, seed=5, residual_ln=(mean, rstd, gam, bet)),
    "dgrad FFN2 (x gelu' tape) N3072 K768": lambda: ops.gemm_nt(x, W1, out_f, T, gelu_pre=pre, act=2),
    "dgrad FFN1 (plain fp16 out) N768 K3072": lambda: ops.gemm_nt(hbig, W2, out_d, T),
    "QKV fwd (bias) N2304 K768": lambda: ops.gemm_nt(x, h16(2304, d) * 0.05, torch.empty(T, 2304, device=dev, dtype=torch.float16), T, bias=torch.randn(2304, device=dev)),
}
for name, fn in cases.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    st = np.ctypeslib.as_array(lib.cldrd_dev_stamps(), shape=(1024, 8)).astype(np.int64).copy()
    st = st[:min(256, (T // 256) * 4)]  # the first round of tiles (one workgroup per CU)
    us = lambda a: a / 100.0            # 100 MHz
    ph = [("prologue", st[:, 1] - st[:, 0]), ("K loop", st[:, 2] - st[:, 1]), ("barrier", st[:, 3] - st[:, 2])] + \
         [(f"chunk {k}", st[:, 4 + k] - st[:, 3 + k]) for k in range(4)]
    tot = st[:, 7] - st[:, 0]
    print(f"{name}: tile {us(np.median(tot)):6.2f} us = " + " | ".join(f"{n} {us(np.median(v)):5.2f}" for n, v in ph), flush=True)
