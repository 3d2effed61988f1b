#!/usr/bin/env python3
"""Dump the captured training step as a DOT file (hipGraphDebugDotPrint through torch's CUDAGraph.debug_dump) and list, for the kernels right
behind the loss, which nodes they depend on.  Usage on the GPU box: python tools/graph_dot.py [out.dot]"""
import os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cldrd_amd.synthetic as syn
from cldrd_amd.encoder import EncoderConfig
from cldrd_amd.models import NwayDualEncoder
from cldrd_amd.trainer import NwayTrainer
from cldrd_amd.trainer import nway_listwise as TL

out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/step_graph.dot"
dev = torch.device("cuda", 0)
cfg = EncoderConfig(arch="distilbert", dropout=0.1, attention_dropout=0.1)
torch.manual_seed(0)
model = NwayDualEncoder(cfg, share_weights=False).to(dev).train()
tr = NwayTrainer(model, loss="kl_div", T=1.0, learning_rate=7e-6, warmup_steps=4000, total_steps=100000)
batch = syn.nway_batch(4680, 8, 32, 30, 128, ragged=False, label_kind="teacher")
batch = {k: ({kk: vv.to(dev) for kk, vv in v.items()} if isinstance(v, dict) else v.to(dev)) for k, v in batch.items()}
orig = torch.cuda.CUDAGraph
class G(orig):
    def __new__(cls, *a, **k):
        g = orig.__new__(cls, *a, **k)
        g.enable_debug_mode()
        return g
torch.cuda.CUDAGraph = G
TL.torch.cuda.CUDAGraph = G
for _ in range(5):
    tr.train_step(batch)
torch.cuda.synchronize()
g = [e["graph"] for e in tr._graphs.values() if e["graph"] is not None][0]
g.debug_dump(out)
txt = open(out).read()
print("dot bytes", len(txt))
nodes = dict(re.findall(r'"?(\w+)"?\s*\[[^\]]*label="([^"]*)"', txt))
edges = re.findall(r'"?(\w+)"?\s*->\s*"?(\w+)"?', txt)
print("nodes", len(nodes), "edges", len(edges))
pred = {}
for a, b in edges:
    pred.setdefault(b, []).append(a)
short = lambda n: re.sub(r"\s+", " ", nodes.get(n, n))[:70]
hits = [n for n, l in nodes.items() if "scale_apply" in l]
print("scale_apply nodes", hits)
succ = {}
for a, b in edges:
    succ.setdefault(a, []).append(b)
for h in hits:
    for c in succ.get(h, []):
        print("child of scale_apply:", c, short(c), "<- preds:", [(p, short(p)) for p in pred.get(c, [])])
# nodes with more than one predecessor near the tail
multi = [(n, pred[n]) for n in pred if len(pred[n]) > 1]
print("nodes with several predecessors:", len(multi))
for n, ps in multi[:40]:
    print("  ", n, short(n), "<-", [short(p) for p in ps])
