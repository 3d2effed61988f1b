import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import cldrd_amd.synthetic as syn
from cldrd_amd import hip_ops as ops
DEV="cuda"
def run(nq, rows, d, dtype, iters):
    Q = torch.from_numpy(syn.normal(21, nq * d).reshape(nq, d).astype(np.float32)).to(DEV).to(dtype)
    P = torch.from_numpy(syn.normal(22, rows * d).reshape(rows, d).astype(np.float32)).to(DEV).to(dtype)
    S = Q.float() @ P.float().T
    thr = torch.quantile(S[:, :4096], 1.0 - 40.0 / 4096, dim=1).contiguous()
    ref = None
    bad = 0
    for it in range(iters):
        res = []
        for tiled in (False, True):
            counts = torch.zeros(nq + 1, dtype=torch.int32, device=DEV)
            cr = torch.full((nq, 2048), -1, dtype=torch.int32, device=DEV)
            cs = torch.zeros(nq, 2048, device=DEV)
            ops.topk_scan_filter(Q, P, thr, counts, cr, cs, tiled=tiled)
            c = counts.cpu().numpy(); crh = cr.cpu().numpy(); csh = cs.cpu().numpy()
            res.append([dict(zip(crh[q, :c[q]].tolist(), csh[q, :c[q]].tolist())) for q in range(nq)])
        if ref is None: ref = res
        for name, a, b in (("stream-vs-first-stream", res[0], ref[0]), ("tiled-vs-first-tiled", res[1], ref[1]), ("stream-vs-tiled", res[0], res[1])):
            for q in range(nq):
                if a[q].keys() != b[q].keys():
                    diff = set(a[q]) ^ set(b[q])
                    bad += 1
                    if bad < 12:
                        print(f"iter {it} {name} q {q}: {len(diff)} differing rows; thr {thr[q].item():.6f};", [(r, a[q].get(r), b[q].get(r), float(S[q, r])) for r in list(diff)[:3]], flush=True)
    print(nq, rows, d, dtype, "iterations", iters, "mismatching (iteration, query) pairs:", bad, flush=True)
run(128, 70001, 256, torch.bfloat16, 40)
run(128, 70001, 256, torch.float16, 40)
run(128, 40000, 768, torch.float16, 20)
run(256, 40000, 768, torch.float16, 20)
