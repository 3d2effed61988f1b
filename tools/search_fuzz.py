#!/usr/bin/env python3
"""Randomised exactness check of FlatIPIndex.search: random shapes (rows 1e3 .. 4e5, d in {128, 256, 768}, 1 .. 700 queries, k in {1, 10, 100,
1000}) x random corpora (isotropic, CLS-like with a common component, duplicated rows, heavy-tailed norms, queries that ARE corpus rows)
against an independent fp64 reference computed on the device.  usage: tools/search_fuzz.py [cases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from cldrd_amd.retriever import retrieval_utils as RU
from test_gpu_retrieval import same_ranking, _device_fp64_topk
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
DEV = "cuda"
bad = 0
t0 = time.time()
for c in range(cases):
    rows = int(10 ** rng.uniform(3, 5.6)); d = int(rng.choice([128, 256, 768])); nq = int(rng.choice([1, 3, 17, 128, 129, 300, 700]))
    k = int(rng.choice([1, 10, 100, 1000])); kind = str(rng.choice(["iso", "cls", "dup", "heavy", "self"]))
    g = torch.Generator(device=DEV).manual_seed(int(rng.integers(1 << 30)))
    P = torch.randn(rows, d, device=DEV, generator=g)
    Q = torch.randn(nq, d, device=DEV, generator=g)
    if kind == "cls":
        common = torch.randn(d, device=DEV, generator=g) * 3.0
        P = common[None] * (0.8 + 0.4 * torch.rand(rows, 1, device=DEV, generator=g)) + 0.3 * P
        Q = common[None] * (0.8 + 0.4 * torch.rand(nq, 1, device=DEV, generator=g)) + 0.3 * Q
    elif kind == "dup":
        P[rows // 2:] = P[: rows - rows // 2].clone()
    elif kind == "heavy":
        P = P * torch.exp(1.5 * torch.randn(rows, 1, device=DEV, generator=g))
    elif kind == "self":
        Q = P[torch.randint(0, rows, (nq,), device=DEV, generator=g)].clone()
    ids = np.arange(rows, dtype=np.int64) * 2 + 1
    index = RU.construct_flatindex_from_embeddings(P.cpu().numpy(), ids)
    RU.convert_index_to_gpu(index, 0, False)
    D, I = index.search(Q.cpu().numpy(), k)
    st = index.last_stats
    sel = np.unique(rng.integers(0, nq, size=min(nq, 24)))
    Dg, Ig = _device_fp64_topk(P, Q[torch.from_numpy(sel).to(DEV)], k)
    Ig = np.where(Ig >= 0, Ig * 2 + 1, -1)
    try:
        swaps = same_ranking(D[sel], I[sel], Dg, Ig)
        ok = "ok"
    except AssertionError as e:
        bad += 1; ok = "MISMATCH " + str(e)[:200]; swaps = -1
    print(f"case {c}: rows {rows} d {d} nq {nq} k {k} {kind}: {ok} (near-tie swaps {swaps}; scans {st.get('scans')}, rescans {st.get('rescans')}, "
          f"exhaustive {st.get('exhaustive')}, fallback {st.get('fallback', 0)})", flush=True)
    del index, P, Q
print(f"{cases} cases, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
