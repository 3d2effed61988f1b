#!/usr/bin/env python3
"""Micro-benchmark of the NT GEMM variants on the encoder shapes (interleaved rounds in one process)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401  (development build: the knobs below do not exist in the product library)
import torch
from cldrd_amd import hip_ops as ops

def main():
    dev = "cuda"
    T = int(os.environ.get("T", 32768))
    # the eight Linear GEMMs of one encoder layer (forward + data gradients) with the epilogues the trainer uses (fp32 residual stream)
    shapes = [("qkv", T, 2304, 768, {"bias": 1}), ("out", T, 768, 768, {"bias": 1, "res32": 1}),
              ("ffn1", T, 3072, 768, {"bias": 1, "pre": 1, "act": 3}), ("ffn1_old", T, 3072, 768, {"bias": 1, "pre": 1, "act": 1}), ("ffn2", T, 768, 3072, {"bias": 1, "res32": 1, "drop": 1}),
              ("dgrad_ffn2", T, 3072, 768, {"gp": 1, "act": 2}), ("dgrad_ffn2_old", T, 3072, 768, {"gp": 1}), ("dgrad_ffn1", T, 768, 3072, {"res": 1}),
              ("dgrad_out", T, 768, 768, {}), ("dgrad_qkv", T, 768, 2304, {"res": 1})]
    variants = os.environ.get("VARIANTS", "ring,ring2s").split(",")      # ring | ring2s | ring128 | ring192 | ring256
    torch.manual_seed(0)
    tot = {v: 0.0 for v in variants}
    for name, M, N, K, ep in shapes:
        # ROTATE=n: n sets of (A, output, residual, ...) used in turn, so that no call finds its operands in the 256-MB Infinity Cache
        # (as in the training step, where every activation is touched once per pass); 1 = the same buffers every call
        rot = int(os.environ.get("ROTATE", 1))
        As = [torch.randn(M, K, device=dev).bfloat16() for _ in range(rot)]
        B = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
        outs = [torch.empty(M, N, device=dev, dtype=torch.float32 if ep.get("res32") else torch.bfloat16) for _ in range(rot)]
        A, out = As[0], outs[0]
        kw = {}
        if ep.get("bias"): kw["bias"] = torch.randn(N, device=dev)
        if ep.get("res"): kw["residual"] = torch.randn(M, N, device=dev).bfloat16()
        if ep.get("res32"): kw["residual"] = torch.randn(M, N, device=dev)
        if ep.get("pre"): kw["preact"] = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        if ep.get("act"): kw["act"] = ep["act"]
        if ep.get("gp"): kw["gelu_pre"] = torch.randn(M, N, device=dev).bfloat16()
        if ep.get("drop"): kw.update(dropout_p=0.1, seed=1234)
        kws = [kw] + [{k: (v.clone() if torch.is_tensor(v) and v.dim() == 2 else v) for k, v in kw.items()} for _ in range(rot - 1)]
        res = {}
        for rnd in range(3):
            for v in variants:
                os.environ["CLDRD_GEMM_ASYM"] = "0" if v == "ring2s" else "1"          # ring2s: two A slots (the round-2 K-loop schedule)
                if v.startswith("ring") and len(v) > 4 and v != "ring2s": os.environ["CLDRD_GEMM_TILE"] = v[4:]
                else: os.environ.pop("CLDRD_GEMM_TILE", None)
                for i in range(2): ops.gemm_nt(As[i % rot], B, outs[i % rot], **kws[i % rot])
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(10): ops.gemm_nt(As[i % rot], B, outs[i % rot], **kws[i % rot])
                e1.record(); torch.cuda.synchronize()
                res.setdefault(v, []).append(e0.elapsed_time(e1) / 10)
        fl = 2.0 * M * N * K
        line = f"{name:11s} M={M} N={N} K={K}: "
        for v, ts in res.items():
            t = min(ts)
            if not name.endswith("_old"): tot[v] += t
            line += f" {v}: {t*1e3:7.1f} us {fl/t/1e9:7.1f} TF/s |"
        print(line, flush=True)
    print("layer total (8 GEMMs): " + " | ".join(f"{v}: {t*1e3:7.1f} us" for v, t in tot.items()), flush=True)
    os.environ.pop("CLDRD_GEMM_TILE", None)
    # weight gradients: dW[N1,N2] = dY[T,N1]^T X[T,N2] (+ bias gradient), split-K slabs + reduction included
    for name, N1, N2 in [("w_qkv", 2304, 768), ("w_out", 768, 768), ("w_ffn1", 3072, 768), ("w_ffn2", 768, 3072)]:
        dY = (torch.randn(T, N1, device=dev) * 0.02).bfloat16()
        X = torch.randn(T, N2, device=dev).bfloat16()
        dW = torch.empty(N1, N2, device=dev)
        db = torch.empty(N1, device=dev)
        ws = torch.empty(ops.wgrad_workspace_elems(T, N1, N2), device=dev)
        ts = []
        for rnd in range(3):
            for _ in range(2): ops.wgrad(dY, X, dW, T, ws, dbias=db)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.wgrad(dY, X, dW, T, ws, dbias=db)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        t = min(ts)
        print(f"{name:11s} T={T} N1={N1} N2={N2}: {t*1e3:7.1f} us {2.0*T*N1*N2/t/1e9:7.1f} TF/s (slab reduction included)", flush=True)

if __name__ == "__main__":
    main()
