"""BERT-large-like widths through the whole step against the oracle (dim 1024, 16 heads, FFN 4096, L = 200 > 128)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import cldrd_amd.synthetic as syn, selftest
from cldrd_amd.encoder import EncoderConfig
from cldrd_amd.trainer import NwayTrainer
from oracle import encoder_ref as E
for arch, L in (("bert", 200), ("distilbert", 96)):
    cfg = EncoderConfig(arch=arch, vocab_size=1000, dim=1024, n_heads=16, hidden_dim=4096, n_layers=2, max_position_embeddings=256, dropout=0.0, attention_dropout=0.0)
    model = selftest.build_tiny_model(cfg, std=0.02).cuda().train()        # the HF init scale
    batch = syn.nway_batch(4680, 2, 3, 12, L, vocab=cfg.vocab_size, ragged=True)
    tr = NwayTrainer(model, loss="margin_mse")
    _, logits = tr.forward_backward(batch)
    qp, pp = selftest.oracle_params(model)
    ref = E.nway_forward(qp, pp, selftest.oracle_cfg(cfg), batch["query"], batch["nway_passages"]).detach().numpy()
    err = np.abs(logits.cpu().numpy() - ref).max() / np.abs(ref).max()
    out = tr.train_step(batch)
    print(arch, "L", L, "logits rel err vs oracle", round(float(err), 5), "loss", float(out[0]), "finite", bool(torch.isfinite(tr.flat_p).all()))
