"""cProfile of the HOST side of a training step (cfg2): where do the ~25 us per C-ABI call go?  Top functions by own time."""
import os, sys, cProfile, pstats, io
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
pr = cProfile.Profile()
pr.enable()
for _ in range(20): tr.train_step(batch)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
