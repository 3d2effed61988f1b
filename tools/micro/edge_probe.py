"""Degenerate shapes through the public surface: B = N = 1, one-token sequences, packed batches of length-1 rows, tiny / empty searches."""
import os, sys, traceback
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import cldrd_amd.synthetic as syn, selftest
from cldrd_amd.encoder import EncoderConfig
from cldrd_amd.trainer import NwayTrainer
from cldrd_amd.retriever import retrieval_utils as RU
from oracle import encoder_ref as E
cfg = EncoderConfig(arch="distilbert", vocab_size=512, dim=128, n_heads=2, hidden_dim=256, n_layers=2, max_position_embeddings=64, dropout=0.1, attention_dropout=0.1)
def run(name, fn):
    try:
        print(name, "->", fn(), flush=True)
    except Exception as e:
        print(name, "-> EXC", type(e).__name__, str(e)[:200], flush=True)
def step(B, N, Lq, Lp, lens=None, loss="margin_mse"):
    model = selftest.build_tiny_model(cfg).cuda().train()
    batch = syn.nway_batch(1, B, N, Lq, Lp, vocab=cfg.vocab_size, ragged=False)
    if lens is not None:
        m = batch["nway_passages"]["attention_mask"]
        for i, l in enumerate(lens): m.view(-1, Lp)[i, l:] = 0
        batch["nway_passages"]["lengths"] = torch.tensor(lens)
    tr = NwayTrainer(model, loss=loss)
    out = tr.train_step(batch); out2 = tr.train_step(batch)
    return [float(out[0]), float(out2[0]), bool(torch.isfinite(tr.flat_p).all())]
run("B=1 N=1", lambda: step(1, 1, 4, 8))
run("B=1 N=2 kl", lambda: step(1, 2, 4, 8, loss="kl_div"))
run("Lq=1 Lp=1", lambda: step(2, 3, 1, 1))
run("packed, all length 1", lambda: step(2, 3, 4, 8, lens=[1] * 6))
run("packed, mixed 1..8", lambda: step(2, 3, 4, 8, lens=[1, 8, 3, 1, 8, 2]))
run("lambda_mrr N=1", lambda: step(2, 1, 4, 8, loss="lambda_mrr"))
emb = syn.corpus_embeddings(3, 50, 128)
def search(nq, k, rows=50):
    index = RU.construct_flatindex_from_embeddings(emb[:rows], np.arange(rows, dtype=np.int64))
    RU.convert_index_to_gpu(index, 0, False)
    D, I = index.search(syn.corpus_embeddings(4, max(nq, 1), 128)[:nq], k)
    return D.shape, I.shape, (I[:, :min(k, rows)] >= 0).all() if nq else True
run("search nq=1 k=1", lambda: search(1, 1))
run("search k > rows", lambda: search(3, 100))
run("search nq=0", lambda: search(0, 10))
run("search rows=1", lambda: search(2, 5, rows=1))
