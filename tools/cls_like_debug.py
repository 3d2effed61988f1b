#!/usr/bin/env python3
"""First-pass status of the flat-IP search on the CLS-like shard: which proof conditions fail, list lengths, band sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cldrd_amd.synthetic as syn
from cldrd_amd.retriever import retrieval_utils as RU
dev = torch.device("cuda")
rows, d, nq, k = 1105228, 768, int(os.environ.get("NQ", 1024)), 1000
P, u = syn.cls_like_corpus(rows, d, 777, dev)
Q = syn.cls_like_queries(nq, u, 778)
idx = RU.FlatIPIndex.from_device_rows(P)
idx.profile = True
# instrument _run to capture the first-pass outputs
orig = idx._run
cap = {}
def spy(q32, qh, thr, eps, k_, ws, D, I, exh, **kw):
    out = orig(q32, qh, thr, eps, k_, ws, D, I, exh, **kw)
    st, cnt, n2, kh = out
    key = len(cap)
    cap[key] = dict(st=st.cpu().numpy().copy(), n2=n2.cpu().numpy().copy(), kh=kh.cpu().numpy().copy(), thr=thr.cpu().numpy().copy(), eps=eps.cpu().numpy().copy(),
                    cnt=cnt.cpu().numpy().copy(), nq=q32.shape[0])
    return out
idx._run = spy
D, I, stats = idx.search_device(Q, k)
torch.cuda.synchronize()
print("stats", {k_: v for k_, v in stats.items()})
for key, c in cap.items():
    st = c["st"]
    bits = {b: int(((st & b) != 0).sum()) for b in (1, 2, 4, 8, 16)}
    QT = idx.query_tile
    nb = (c["nq"] + QT - 1) // QT
    lens = c["cnt"].reshape(nb, QT + 1)[:, :QT].reshape(-1)[:c["nq"]]
    gap = (c["kh"] - c["thr"]) / c["eps"]
    print(f"run {key}: nq {c['nq']} status bits {bits}; list length min/med/max {lens.min()}/{int(np.median(lens))}/{lens.max()}; kept n2 min/med/max {c['n2'].min()}/{int(np.median(c['n2']))}/{c['n2'].max()}; "
          f"(t^ - thr)/eps min/med/max {np.nanmin(gap):.2f}/{np.nanmedian(gap):.2f}/{np.nanmax(gap):.2f}; eps med {np.median(c['eps']):.4f}")
# density of scores near the k-th for one query (fp32)
s = (P @ Q[0]).sort(descending=True).values
kth = float(s[k - 1])
e0 = float(cap[0]["eps"][0])
print(f"query 0: k-th score {kth:.4f}, eps {e0:.4f}, rows within 2 eps below the k-th: {int(((s < kth) & (s >= kth - 2 * e0)).sum())}, within 4 eps: {int(((s < kth) & (s >= kth - 4 * e0)).sum())}, "
      f"|q| {float(Q[0].norm()):.3f} max|p| {idx._max_norm:.3f}, |p| of the top rows {float(P[(P @ Q[0]).topk(5).indices].norm(dim=1).mean()):.3f}")
