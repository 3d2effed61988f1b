import os, sys, time, json
sys.path.insert(0, os.getcwd())
import torch, torch.distributed as dist
mode = sys.argv[1]
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
if mode == "nccl":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
elif mode == "nccl_lazy":
    dist.init_process_group("nccl", rank=0, world_size=1)
elif mode == "gloo":
    dist.init_process_group("gloo", rank=0, world_size=1)
import cldrd_amd.synthetic as syn
from cldrd_amd.encoder import EncoderConfig
from cldrd_amd.models import NwayDualEncoder
from cldrd_amd.trainer import NwayTrainer
cfg = EncoderConfig(arch="distilbert", dropout=0.1, attention_dropout=0.1)
torch.manual_seed(0)
model = NwayDualEncoder(cfg, share_weights=False).to(dev); model.train()
tr = NwayTrainer(model, loss="kl_div", T=1.0, learning_rate=7e-6, warmup_steps=4000, total_steps=100000)
batch = syn.nway_batch(4680, 8, 32, 30, 128, ragged=False, label_kind="teacher")
batch = {k: ({kk: vv.to(dev) for kk, vv in v.items()} if isinstance(v, dict) else v.to(dev)) for k, v in batch.items()}
for _ in range(8): tr.train_step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
th = 0.0
for _ in range(30):
    h0 = time.perf_counter(); tr.train_step(batch); th += time.perf_counter() - h0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"mode": mode, "distributed": tr.distributed, "ms_per_step": round(1e3 * dt / 30, 3), "host_ms_per_step": round(1e3 * th / 30, 3)}))
if mode != "none": dist.destroy_process_group()
