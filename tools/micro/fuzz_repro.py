import os, sys
ROOT = os.getcwd(); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from cldrd_amd.retriever import retrieval_utils as RU
from test_gpu_retrieval import _device_fp64_topk
DEV = "cuda"
for seed in range(6):
    for rows, d, nq, k in ((2873, 128, 128, 1000), (4541, 768, 128, 1000), (9000, 128, 64, 1000)):
        g = torch.Generator(device=DEV).manual_seed(seed)
        P = torch.randn(rows, d, device=DEV, generator=g) * torch.exp(1.5 * torch.randn(rows, 1, device=DEV, generator=g))
        Q = torch.randn(nq, d, device=DEV, generator=g)
        index = RU.construct_flatindex_from_embeddings(P.cpu().numpy(), np.arange(rows, dtype=np.int64))
        RU.convert_index_to_gpu(index, 0, False)
        D, I = index.search(Q.cpu().numpy(), k)
        Dg, Ig = _device_fp64_topk(P, Q, k)
        fin = np.isfinite(Dg)
        rel = np.abs(D[fin] - Dg[fin]) / (np.abs(Dg[fin]) + 1e-6)
        worst = np.unravel_index(np.argmax(np.where(fin, np.abs(D - Dg) / (np.abs(Dg) + 1e-6), 0)), D.shape)
        ids_equal = np.mean(I == Ig)
        print(f"seed {seed} rows {rows} d {d}: finite same {np.array_equal(np.isfinite(D), fin)}, max rel score err {rel.max():.2e} at {worst} (D {D[worst]:.6f} ref {Dg[worst]:.6f} id {I[worst]} ref id {Ig[worst]}), ids equal {ids_equal:.4f}, exhaustive {index.last_stats['exhaustive']}", flush=True)
