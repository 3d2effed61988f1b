import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, numpy as np
import cldrd_amd.synthetic as syn, selftest
from cldrd_amd.encoder import EncoderConfig
from cldrd_amd.trainer import NwayTrainer
os.environ["CLDRD_GRAPH"] = "0"
for arch in ("bert", "distilbert"):
    cfg = EncoderConfig(arch=arch, vocab_size=512, dim=128, n_heads=2, hidden_dim=256, n_layers=3, max_position_embeddings=64, dropout=0.0, attention_dropout=0.0)
    batch = syn.nway_batch(4680, 3, 4, 10, 32, vocab=cfg.vocab_size, ragged=True)
    out = {}
    for v in ("1", "0"):
        os.environ["CLDRD_LN_ON_THE_FLY"] = v
        model = selftest.build_tiny_model(cfg).cuda().train()
        tr = NwayTrainer(model, loss="margin_mse")
        _, lg = tr.forward_backward(batch)
        out[v] = (lg.double().clone(), tr.flat_g.double().clone())
    d = (out["1"][0] - out["0"][0]).abs().max().item() / out["1"][0].abs().max().item()
    g = ((out["1"][1] - out["0"][1]).norm() / out["1"][1].norm()).item()
    print(arch, "logits rel diff", d, "grad rel diff", g)
