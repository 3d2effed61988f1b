#!/usr/bin/env python3
"""Logits / loss of the cfg2 golden model with the fp32 LayerNorm output stored (CLDRD_LN_ON_THE_FLY=0) and applied on the fly (default)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_model as T
from cldrd_amd.trainer.nway_listwise import NwayTrainer
import cldrd_amd.synthetic as syn
g = np.load(os.path.join(T.GOLDEN, "full_distilbert_cfg2.npz"))
B, N, Lq, Lp = int(g["B"]), int(g["N"]), int(g["Lq"]), int(g["Lp"])
batch = syn.nway_batch(4680, B, N, Lq, Lp, ragged=True, label_kind=str(g["label_kind"]) if "label_kind" in g.files else "teacher")
ref = g["logits"]
out = {}
for mode in ("0", "1"):
    os.environ["CLDRD_LN_ON_THE_FLY"] = mode
    model = T._full_size_model("distilbert", 6)
    tr = NwayTrainer(model, loss="kl_div")
    loss, logits = tr.forward_backward(batch)
    out[mode] = logits.cpu().numpy()
    print(f"on_the_fly={mode}: loss {loss[0].item():.5f} (reference {float(g['loss_kl']):.5f}, autocast {float(g['loss_kl_autocast']):.5f}); max|dlogit| vs ref {np.abs(out[mode]-ref).max():.4f} mean signed {np.mean(out[mode]-ref):+.4f}")
print("between the two paths: max|d| %.5f rms %.5f" % (np.abs(out["0"] - out["1"]).max(), np.sqrt(np.mean((out["0"] - out["1"]) ** 2))))
