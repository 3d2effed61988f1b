import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import cldrd_amd.synthetic as syn
from cldrd_amd import hip_ops as ops
from cldrd_amd.retriever import retrieval_utils as RU
from oracle import retrieval_ref as R
n, d, nq, k = 20000, 768, 150, 100
emb = syn.corpus_embeddings(11, n, d); emb[n // 2] = emb[n // 3]
q = syn.corpus_embeddings(12, nq, d); q[0] = emb[17] * 1.0
index = RU.construct_flatindex_from_embeddings(emb, None)
RU.convert_index_to_gpu(index, 0, False)
for QT in (256, 128):
    index.query_tile = QT; index._ws = None
    D, I = index.search(q, k)
    Dr, Ir = R.flat_ip_search(emb, None, q, k)
    badq = [i for i in range(nq) if not np.array_equal(np.sort(I[i]), np.sort(Ir[i]))]
    print("QT", QT, "stats", index.last_stats, "queries with wrong id sets:", badq[:20])
    for i in badq[:3]:
        miss = set(Ir[i].tolist()) - set(I[i].tolist())
        print("  query", i, "missing rows", miss, "ref scores", [float(Dr[i][list(Ir[i]).index(r)]) for r in miss], "kth ref", float(Dr[i, -1]), "ours kth", float(D[i, -1]))
# low level on the failing tile
dev = torch.device("cuda")
q32 = torch.from_numpy(q).to(dev)
qh = torch.empty(nq, d, dtype=torch.float16, device=dev); qb = torch.empty(nq, d, dtype=torch.bfloat16, device=dev)
qn = torch.empty(nq, device=dev); flag = torch.zeros(1, dtype=torch.int32, device=dev)
ops.topk_prep_queries(q32, qh, qb, qn, flag)
S16 = (qh.float() @ index._p16.float().T)
S32 = torch.from_numpy(q.astype(np.float64) @ emb.astype(np.float64).T).to(dev)
eps = torch.empty(nq, device=dev); ops.topk_thresholds(None, qn, index._max_norm, d, None, eps)
print("max |fp16 score - exact| / eps over all pairs:", float(((S16.double() - S32).abs() / eps.double()[:, None]).max()))
for thr_v in (9.0,):
    for m in (150, 256 if False else 128):
        mm = min(m, nq)
        thr = torch.full((mm,), thr_v, device=dev)
        counts = torch.zeros(mm + 1, dtype=torch.int32, device=dev)
        cr = torch.full((mm, 8192), -1, dtype=torch.int32, device=dev); cs = torch.zeros(mm, 8192, device=dev)
        ops.topk_scan_filter(qh[:mm].contiguous(), index._p16, thr, counts, cr, cs)
        c = counts.cpu().numpy()
        exp = (S16[:mm] >= thr_v).sum(1).cpu().numpy()
        print("scan nq", mm, "thr", thr_v, "count mismatches vs torch fp16-operand scores:", int((np.abs(c[:mm] - exp) > 2).sum()), "max diff", int(np.abs(c[:mm] - exp).max()), "dropped", c[mm])
