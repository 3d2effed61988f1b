#!/usr/bin/env python3
"""Small-M NT GEMMs (the CLS-only tail of the passage tower, the query tower: M = 256 rows): one-launch 64 x 64 kernel against the
128 x 128 kernel with split-K + finishing launch.  Each shape is timed as a HIP graph of 40 dependent launches (the step replays a graph:
launch overhead of the host is not what the step pays).  Usage: python tools/small_gemm_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cldrd_amd import hip_ops as ops


def graph_time(fn, n=40, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * reps)


def main():
    dev = "cuda"
    M = int(os.environ.get("M", 256))
    shapes = [("q / out-proj / dctx", 768, 768, {"bias": 1}), ("out-proj + fp32 residual", 768, 768, {"bias": 1, "res32": 1}),
              ("qkv", 2304, 768, {"bias": 1}), ("ffn1 + gelu tape", 3072, 768, {"bias": 1, "pre": 1, "act": 3}),
              ("dgrad ffn2 (gelu')", 3072, 768, {"gp": 1, "act": 2}), ("dgrad qkv", 768, 2304, {}), ("ffn2", 768, 3072, {"bias": 1, "res32": 1}),
              ("dgrad ffn1", 768, 3072, {})]
    torch.manual_seed(0)
    for name, N, K, ep in shapes:
        A = torch.randn(M, K, device=dev).half()
        B = (torch.randn(N, K, device=dev) * 0.02).half()
        out = torch.empty(M, N, device=dev, dtype=torch.float32 if ep.get("res32") else torch.float16)
        kw = {}
        if ep.get("bias"): kw["bias"] = torch.randn(N, device=dev)
        if ep.get("res32"): kw["residual"] = torch.randn(M, N, device=dev)
        if ep.get("pre"): kw["preact"] = torch.empty(M, N, device=dev, dtype=torch.float16)
        if ep.get("act"): kw["act"] = ep["act"]
        if ep.get("gp"): kw["gelu_pre"] = torch.rand(M, N, device=dev).half()
        res = {}
        for nt64 in (1, 0):
            ops.set_tuning("gemm_nt64", nt64)
            try:
                res[nt64] = graph_time(lambda: ops.gemm_nt(A, B, out, **kw))
            finally:
                ops.set_tuning("gemm_nt64", 1)
        print(f"{name:26s} M={M} N={N:5d} K={K:5d}: 64x64 {res[1]:6.2f} us | 128x128 (+ split-K) {res[0]:6.2f} us", flush=True)


if __name__ == "__main__":
    main()
