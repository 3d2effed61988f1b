import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import cldrd_amd.synthetic as syn, selftest
from cldrd_amd.encoder import EncoderConfig
from oracle import encoder_ref as E
cfg = EncoderConfig(arch="distilbert", vocab_size=512, dim=128, n_heads=2, hidden_dim=256, n_layers=2, max_position_embeddings=64, dropout=0.0, attention_dropout=0.0)
for cls_only in (True, False):
    for L in (1, 2, 3, 5, 33):
        model = selftest.build_tiny_model(cfg).cuda().eval()
        for t in model.towers(): t.cls_only_last = cls_only
        seq = syn.seq_batch(7, 4, L, vocab=cfg.vocab_size)["seq"]
        got = model.passage_embs({k: v.cuda() for k, v in seq.items()}).detach().float().cpu().numpy()
        qp, pp = selftest.oracle_params(model)
        ref = E.cls_embs(pp, selftest.oracle_cfg(cfg), seq).detach().numpy()
        print("cls_only", cls_only, "L", L, "max rel err", float(np.abs(got - ref).max() / np.abs(ref).max()))
