#!/usr/bin/env python3
"""The retrieve leg alone (for rocprofv3 --kernel-trace --stats): cfg5 shard, 6980 queries, k = 1000, device-resident search x3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cldrd_amd.retriever.retrieval_utils import FlatIPIndex
dev = torch.device("cuda", 0)
rows, D, nq = 1105228, 768, 6980
gen = torch.Generator(device=dev).manual_seed(1234)
P = torch.randn(rows, D, device=dev, generator=gen)
P *= ((9.0 + 3.0 * torch.rand(rows, 1, device=dev, generator=gen)) / P.norm(dim=1, keepdim=True))
idx = FlatIPIndex.from_device_rows(P)
q = torch.randn(nq, D, device=dev, generator=gen)
q *= 10.0 / q.norm(dim=1, keepdim=True)
idx.profile = True
for _ in range(3):
    _, _, st = idx.search_device(q, 1000)
torch.cuda.synchronize()
print(st)
