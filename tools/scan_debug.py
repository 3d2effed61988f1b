import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from cldrd_amd import hip_ops as ops
dev = "cuda"
for rows in (20000, 40000, 8200):
    d = 768
    g = torch.Generator(device=dev).manual_seed(5)
    P = torch.randn(rows, d, device=dev, generator=g); P *= 10.5 / P.norm(dim=1, keepdim=True)
    Q = torch.randn(256, d, device=dev, generator=g); Q *= 10.0 / Q.norm(dim=1, keepdim=True)
    Ph, Qh = P.half(), Q.half()
    S = Qh.float() @ Ph.float().T
    def scan(nq, thr_v):
        thr = torch.full((nq,), thr_v, device=dev)
        counts = torch.zeros(nq + 1, dtype=torch.int32, device=dev)
        cr = torch.full((nq, 8192), -1, dtype=torch.int32, device=dev); cs = torch.zeros(nq, 8192, device=dev)
        ops.topk_scan_filter(Qh[:nq].contiguous(), Ph, thr, counts, cr, cs)
        c = counts.cpu().numpy(); crh = cr.cpu().numpy()
        return [set(crh[q, :c[q]].tolist()) for q in range(nq)], c
    for thr_v in (9.0, 11.0):
        a, ca = scan(128, thr_v)
        b, cb = scan(129, thr_v)
        exp = [set(torch.nonzero(S[q] >= thr_v).flatten().tolist()) for q in range(128)]
        miss_a = [(q, r) for q in range(128) for r in exp[q] - a[q]]
        miss_b = [(q, r) for q in range(128) for r in exp[q] - b[q]]
        extra_a = sum(len(a[q] - exp[q]) for q in range(128))
        print(f"rows {rows} thr {thr_v}: expected hits {sum(len(e) for e in exp)}; <8,1> missing {len(miss_a)} extra {extra_a}; <8,2> missing {len(miss_b)}")
        if miss_a:
            rr = np.array([r for _, r in miss_a]); qq = np.array([q for q, _ in miss_a])
            print("   missing rows: tile index mod 256 (workgroup):", np.unique((rr // 32) % 256)[:20], "tile round:", np.unique(rr // 32 // 256), "row in tile:", np.unique(rr % 32),
                  "queries mod 16:", np.unique(qq % 16), "query // 16 (wave):", np.unique(qq // 16))
            for q, r in miss_a[:5]:
                print("     q", q, "row", r, "score", float(S[q, r]))
