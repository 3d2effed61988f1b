#!/usr/bin/env python3
"""Step-by-step comparison of the HIP-graph replay of NwayTrainer.train_step with the eager step (two trainers on identical models):
reports, per step, the first buffer (loss, gradients, parameters, Adam moments) that differs and which parameter it belongs to."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cldrd_amd.synthetic as syn
from cldrd_amd.encoder import EncoderConfig
from cldrd_amd.models import NwayDualEncoder
from cldrd_amd.trainer import NwayTrainer

drop = float(os.environ.get("DROP", "0.1"))
cfg = EncoderConfig(arch="distilbert", vocab_size=512, dim=128, n_heads=2, hidden_dim=256, n_layers=2, max_position_embeddings=64,
                    dropout=drop, attention_dropout=drop)


def make():
    torch.manual_seed(0)
    model = NwayDualEncoder(cfg, share_weights=False).cuda().train()
    with torch.no_grad():
        for seed, tower in ((11, model.query_encoder), (12, model.passage_encoder)):
            for name, p in tower.named_flat():
                p.copy_(syn.init_param(seed, name, tuple(p.shape), std=0.05, perturb=True))
    return NwayTrainer(model, loss="kl_div", learning_rate=3e-3, warmup_steps=5, total_steps=40)


def where(tr, idx):
    for ti, (tower, toff) in enumerate(zip(tr.model.towers(), tr.model._tower_offsets)):
        for n in tower.layout.order:
            off, shape = tower.layout.entries[n]
            num = 1
            for s in shape:
                num *= s
            if toff + off <= idx < toff + off + num:
                return f"tower{ti}.{n}[{idx - toff - off}]"
    return "?"


a, b = make(), make()
for i in range(8):
    batch = syn.nway_batch(100 + i, 3, 4, 8, 16, vocab=cfg.vocab_size, ragged=True, label_kind="teacher")
    batch = {k: ({kk: vv.cuda() for kk, vv in v.items()} if isinstance(v, dict) else v.cuda()) for k, v in batch.items()}
    os.environ["CLDRD_GRAPH"] = "1"
    la = a.train_step(batch).clone()
    os.environ["CLDRD_GRAPH"] = "0"
    lb = b.train_step(batch).clone()
    torch.cuda.synchronize()
    used = bool(getattr(a, "_graphs", None)) and any(e["graph"] is not None for e in a._graphs.values())
    msg = [f"step {i}: graph={used} loss {la[0].item():.6f} / {lb[0].item():.6f}"]
    for name, x, y in (("logits", a.last_logits, b.last_logits), ("flat_g", a.flat_g, b.flat_g), ("clip", a.clip, b.clip), ("flat_p", a.flat_p, b.flat_p),
                       ("m", a.m, b.m), ("v", a.v, b.v), ("shadow", a._shadow.float(), b._shadow.float())):
        if not torch.equal(x, y):
            d = (x.float() - y.float()).abs().view(-1)
            j = int(d.argmax())
            first = int((d != 0).nonzero()[0])
            msg.append(f"  {name}: {int((d != 0).sum())} of {d.numel()} differ, max {d.max().item():.3e} at {j}"
                       + (f" ({where(a, j)}); first at {first} ({where(a, first)})" if name in ("flat_g", "flat_p", "m", "v", "shadow") else f" values {x.view(-1)[:3].tolist()} / {y.view(-1)[:3].tolist()}"))
    print("\n".join(msg), flush=True)
