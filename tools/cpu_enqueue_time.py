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
for _ in range(5): tr.train_step(batch)
torch.cuda.synchronize()
enq = []
t_all0 = time.perf_counter()
for _ in range(20):
    t0 = time.perf_counter(); tr.train_step(batch); enq.append(time.perf_counter() - t0)
torch.cuda.synchronize()
wall = (time.perf_counter() - t_all0) / 20
print(f"enqueue (CPU) per step: median {sorted(enq)[10]*1e3:.2f} ms, min {min(enq)*1e3:.2f} ms; wall per step {wall*1e3:.2f} ms")
