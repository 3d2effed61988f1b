#!/usr/bin/env python3
"""Where a K tile of the weight-gradient (TN) kernel goes: the 256 x 192 kernel on the encoder's shapes with parts of the K loop removed
(development build, CLDRD_TN_ABLATE: 1 no LDS-DMA in the K loop, 2 no fragment reads, 3 no barrier / vmcnt wait; results of modes != 0
are wrong).  One process per mode."""
import os, subprocess, sys
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _devlib  # noqa: F401
    import torch
    from cldrd_amd import hip_ops as ops
    out = []
    for T, N1, N2 in ((4096, 4096, 3072), (32768, 3072, 768), (32768, 768, 3072), (32768, 2304, 768)):
        dY = (torch.rand(T, N1, device="cuda") * 2 - 1).bfloat16(); X = (torch.rand(T, N2, device="cuda") * 2 - 1).bfloat16()
        dW = torch.empty(N1, N2, device="cuda")
        ws = torch.empty(max(1, ops.wgrad_workspace_elems(T, N1, N2)), device="cuda")
        for _ in range(3): ops.wgrad(dY, X, dW, T, ws)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.wgrad(dY, X, dW, T, ws)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        err = ""
        if sys.argv[1] == "0" and T == 4096:
            ref = dY.float().T @ X.float()
            err = f" (max rel err vs fp32 matmul {float((dW - ref).abs().max() / ref.abs().max()):.1e})"
        out.append(f"{T}x{N1}x{N2}: {best*1e3:6.1f} us {2.0*T*N1*N2/best/1e9:5.0f} TF/s-eq{err}")
    print(f"mode {sys.argv[1]}: " + " | ".join(out))
else:
    names = {0: "full kernel", 1: "no LDS-DMA in the K loop", 2: "no fragment reads in the K loop", 3: "no barrier, no vmcnt wait"}
    combos = ((192, 0), (192, 1), (192, 2), (192, 3), (128, 0), (128, 1), (128, 2), (128, 3))
    if os.environ.get("TN_TILES"):          # e.g. TN_TILES=192,256 : the full kernel only, these tiles
        combos = tuple((int(t), 0) for t in os.environ["TN_TILES"].split(","))
    for tile, mode in combos:
        env = dict(os.environ, CLDRD_TN_ABLATE=str(mode), CLDRD_WGRAD_TILE=str(tile))
        r = subprocess.run([sys.executable, __file__, str(mode)], env=env, capture_output=True, text=True)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], "  <-", f"256 x {tile}:", names[mode], flush=True)
