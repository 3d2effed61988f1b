# same-box A/B of the retrieve leg under environment variants: ms per 128-query batch of the enqueued search (HIP events), 5 searches
# after 2 warm-up searches; usage: bash tools/ab_retrieve.sh "A=1" "CLDRD_LIB=/root/repo/ab/x.so" ...
run() { env $1 python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
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
ms = []
for i in range(7):
    _, _, st = idx.search_device(q, 1000)
    if i >= 2: ms.append(st["search_ms"] / 55)
print(" ".join(f"{m:.4f}" for m in ms), "ms per 128-query batch")
PY
}
for rep in 1 2; do
  for v in "$@"; do echo -n "[$rep] ${v:-default}: "; run "${v:-X=1}" 2>/dev/null | tail -1; done
done
