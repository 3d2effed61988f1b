"""Host cost of enqueueing one training step (cfg2): wall time of NwayTrainer.train_step on the CPU (launches are asynchronous) next to
the GPU time per step, and the number of C-ABI calls per step (torch's own fills / copies come on top: see the rocprofv3 summary)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import cldrd_amd.synthetic as syn
from cldrd_amd.encoder import EncoderConfig
from cldrd_amd.models import NwayDualEncoder
from cldrd_amd.trainer import NwayTrainer
dev = torch.device("cuda")
model = NwayDualEncoder(EncoderConfig(arch="distilbert"), share_weights=False).to(dev).train()
tr = NwayTrainer(model, loss="kl_div")
batch = syn.nway_batch(4680, 8, 32, 30, 128, ragged=False, label_kind="teacher")
batch = {k: ({kk: vv.to(dev) for kk, vv in v.items()} if isinstance(v, dict) else v.to(dev)) for k, v in batch.items()}
from cldrd_amd import _lib, hip_ops
ncalls = [0]
_orig = _lib.call
def _counting(name, *a):
    ncalls[0] += 1
    return _orig(name, *a)
hip_ops.call = _counting
for _ in range(5): tr.train_step(batch)
torch.cuda.synchronize()
ncalls[0] = 0
tr.train_step(batch)
torch.cuda.synchronize()
print(f"C-ABI calls per training step: {ncalls[0]}")
hip_ops.call = _orig
# (1) CPU time with an EMPTY queue in front (sync before every step): the pure host cost of enqueueing a step
solo = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); tr.train_step(batch); solo.append(time.perf_counter() - t0)
torch.cuda.synchronize()
print(f"enqueue (CPU, empty queue) per step: median {sorted(solo)[5]*1e3:.2f} ms, min {min(solo)*1e3:.2f} ms")
enq = []
t_all0 = time.perf_counter()
for _ in range(20):
    t0 = time.perf_counter(); tr.train_step(batch); enq.append(time.perf_counter() - t0)
torch.cuda.synchronize()
wall = (time.perf_counter() - t_all0) / 20
print(f"enqueue (CPU) per step: median {sorted(enq)[10]*1e3:.2f} ms, min {min(enq)*1e3:.2f} ms; wall per step {wall*1e3:.2f} ms")
