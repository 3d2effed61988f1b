#!/usr/bin/env python3
"""Where a K tile of the ring GEMM goes: time of the plain BN = 192 kernel at M = 32768, N = 768, K = 3072 with parts of the
K loop removed (CLDRD_GEMM_ABLATE, one process per mode: the mode is latched at first use).  Results of modes != 0 are wrong."""
import os, subprocess, sys
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _devlib  # noqa: F401  (the ablation modes exist in the development build only)
    import torch
    from cldrd_amd import hip_ops as ops
    M, N, K = (int(os.environ.get(k, d)) for k, d in (("M", 32768), ("N", 768), ("K", 3072)))
    A = torch.randn(M, K, device="cuda").bfloat16(); B = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    os.environ["CLDRD_GEMM_TILE"] = "192"
    best = 1e9
    for _ in range(3):
        for _ in range(3): ops.gemm_nt(A, B, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.gemm_nt(A, B, out)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    print(f"mode {sys.argv[1]}: {best*1e3:7.1f} us  {2.0*M*N*K/best/1e9:7.1f} TF/s-equivalent")
else:
    names = {0: "full kernel", 1: "no s_barrier", 2: "no LDS-DMA in the K loop", 3: "no DMA, no barrier, no vmcnt waits", 4: "no fragment reads in the K loop", 6: "DMA issued, no vmcnt wait",
             7: "no epilogue", 8: "epilogue, every tile stored to the first tile's place"}
    for mode in (0, 1, 2, 3, 4, 6, 7, 8):
        env = dict(os.environ, CLDRD_GEMM_ABLATE=str(mode))
        r = subprocess.run([sys.executable, __file__, str(mode)], env=env, capture_output=True, text=True)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], "  <-", names[mode])
