#!/usr/bin/env python3
"""Step time of every training config of BASELINE.json on one GPU (synthetic batches of the config's shape, random init, dropout 0.1, graph replay
where the shape allows): cfg1 DistilBERT B4 N8 L128 margin_mse, cfg2 B8 N32 kl_div (the bench headline), cfg3 B4 N200 margin_mse, cfg4 BERT-base
B4 N64 L256 ranknet.  The parity of these configs is pinned by tests/test_gpu_model.py; this prints what they cost.
`python tools/time_configs.py ragged`: the same configs on MS MARCO-shaped passages with their token counts (packed batches, eager steps)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cldrd_amd.synthetic as syn
from cldrd_amd.encoder import EncoderConfig
from cldrd_amd.models import NwayDualEncoder
from cldrd_amd.trainer import NwayTrainer
dev = torch.device("cuda", 0)
ragged = "ragged" in sys.argv
cfgs = [("cfg1", "distilbert", 4, 8, 30, 128, "margin_mse"), ("cfg2", "distilbert", 8, 32, 30, 128, "kl_div"),
        ("cfg3", "distilbert", 4, 200, 30, 128, "margin_mse"), ("cfg4", "bert", 4, 64, 30, 256, "ranknet")]
only = [a for a in sys.argv[1:] if a.startswith("cfg")]
for name, arch, B, N, Lq, L, loss in cfgs:
    if only and name not in only:
        continue
    cfg = EncoderConfig(arch=arch, dropout=0.1, attention_dropout=0.1) if arch == "distilbert" else \
        EncoderConfig(arch="bert", n_layers=12, dropout=0.1, attention_dropout=0.1, max_position_embeddings=512, vocab_size=30522)
    torch.manual_seed(0)
    model = NwayDualEncoder(cfg, share_weights=False).to(dev).train()
    tr = NwayTrainer(model, loss=loss, T=1.0, learning_rate=7e-6, warmup_steps=4000, total_steps=100000)
    batch = syn.nway_batch(4680, B, N, Lq, L, ragged=ragged, label_kind="teacher" if loss in ("margin_mse", "kl_div") else "mode9")
    fill = float(batch["nway_passages"]["attention_mask"].float().mean())
    if ragged:
        from cldrd_amd.trainer.nway_listwise import batch_to_device
        batch = batch_to_device(batch, dev)
    else:
        batch = {k: ({kk: vv.to(dev) for kk, vv in v.items()} if isinstance(v, dict) else v.to(dev)) for k, v in batch.items()}
    for _ in range(8):
        tr.train_step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 30
    for _ in range(steps):
        tr.train_step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tokens = int(B * N * L * fill)
    replay = any(e["graph"] is not None for e in getattr(tr, "_graphs", {}).values())
    print(f"{name}: {arch} B={B} N={N} L={L} {loss}: {dt * 1e3:8.3f} ms/step  {B / dt:8.1f} samples/s  {tokens / dt / 1e6:6.2f} M passage tokens/s  "
          f"({'graph replay' if replay else 'eager'}{f', packed, token fill {fill:.2f}' if ragged else ''})", flush=True)
    del tr, model
    torch.cuda.empty_cache()
