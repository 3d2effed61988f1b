#!/usr/bin/env python3
"""Full-size goldens of BASELINE.json configs[1..3] produced by IMPORTING THE REFERENCE (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_full_golden.py [cfg2] [cfg3] [cfg4]

What runs is the reference's ``models/nway_dual_encoder.py`` (``NwayDualEncoder.forward`` over HF ``AutoModel``) and
``losses/*.py``; inputs and weights come from the portable generator in ``cl-drd_amd/synthetic.py`` so only OUTPUTS are
stored (logits, CLS vectors, loss, per-tensor gradient L2 norms, and the same quantities under the reference's own
bf16 autocast, which is the drift the reference itself accepts - it trains under autocast, nway_listwise_1.py:334).

Memory: the full-batch forward runs under ``no_grad`` (this is the stored ``logits``); the backward is taken one query
(= N passages) at a time through the reference's own ``forward`` with the matching rows of d loss / d logits - the
N-way logits of sample b depend on sample b only (no in-batch negatives), so the per-sample gradients sum to the
full-batch gradient.
"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (sets sys.path for cldrd_amd, dont_write_bytecode)

syn = MG.syn

FULL = {"distilbert": dict(vocab_size=30522, dim=768, n_heads=12, hidden_dim=3072, n_layers=6, max_position_embeddings=512),
        "bert": dict(vocab_size=30522, dim=768, n_heads=12, hidden_dim=3072, n_layers=12, max_position_embeddings=512)}

# name -> (arch, B, N, Lq, Lp, label_kind, [loss kinds])       SURVEY.md section 8d "Configs as concrete runs"
CONFIGS = {
    "cfg2": ("distilbert", 8, 32, 30, 128, "teacher", ["kl"]),
    "cfg3": ("distilbert", 4, 200, 30, 128, "teacher", ["mse"]),
    "cfg4": ("bert", 4, 64, 30, 256, "mode9", ["ranknet", "lambda"]),
}


def loss_fn(ref_losses, kind):
    return {"kl": lambda yp, yt: ref_losses.KLDiv(1.0)(yp, yt),
            "mse": lambda yp, yt: ref_losses.MarginMSE()(yp, yt),
            "ranknet": lambda yp, yt: ref_losses.ranknet_loss(yp, yt),
            "lambda": lambda yp, yt: ref_losses.lambda_mrr_loss(yp, yt)}[kind]


def sub_batch(batch, b):
    return ({k: v[b:b + 1] for k, v in batch["query"].items()}, {k: v[b:b + 1] for k, v in batch["nway_passages"].items()})


def grads_per_sample(model, batch, dlogits, autocast):
    """Parameter gradients of sum(logits * dlogits) through the reference forward, one query at a time."""
    model.zero_grad()
    B = dlogits.shape[0]
    for b in range(B):
        q, p = sub_batch(batch, b)
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            lg = model(q, p)
        lg.backward(dlogits[b:b + 1].to(lg.dtype))
    gn = {}
    for tower, enc in (("query_encoder", model.query_encoder), ("passage_encoder", model.passage_encoder)):
        for k, p in enc.named_parameters():
            if p.grad is not None:
                gn[f"{tower}.{k}"] = float(p.grad.double().norm())
    return gn


def make(name, NwayDualEncoder, ref_losses):
    arch, B, N, Lq, Lp, label_kind, kinds = CONFIGS[name]
    cfgd = FULL[arch]
    t0 = time.time()
    with tempfile.TemporaryDirectory() as tmp:
        qdir, pdir = os.path.join(tmp, "q"), os.path.join(tmp, "p")
        MG.hf_model(arch, cfgd, seed=11).save_pretrained(qdir)
        MG.hf_model(arch, cfgd, seed=12).save_pretrained(pdir)
        model = NwayDualEncoder(qdir, share_weights=False)
        from transformers import AutoModel
        model.passage_encoder = AutoModel.from_pretrained(pdir)     # the reference loads both towers from one path
    model.eval()                                                     # dropout off: parity is defined at p = 0
    batch = syn.nway_batch(4680, B, N, Lq, Lp, vocab=cfgd["vocab_size"], ragged=True, label_kind=label_kind)
    labels = batch["labels"]
    blob = {"arch": np.array(arch), "B": B, "N": N, "Lq": Lq, "Lp": Lp, "ragged": True, "label_kind": np.array(label_kind),
            "loss_kinds": np.array(kinds)}
    with torch.no_grad():
        logits = model(batch["query"], batch["nway_passages"])
        blob["q_cls"] = model.query_embs(batch["query"]).numpy()
        blob["p_cls"] = model.nway_passage_embs(batch["nway_passages"]).numpy()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            logits_amp = model(batch["query"], batch["nway_passages"]).float()
    blob["logits"], blob["logits_autocast_bf16"] = logits.numpy(), logits_amp.numpy()
    print(f"  {name}: forward done {time.time() - t0:.0f}s; max|logit| {logits.abs().max():.3f}, "
          f"autocast drift {(logits_amp - logits).abs().max():.4f}", flush=True)
    for kind in kinds:
        fn = loss_fn(ref_losses, kind)
        for tag, lg in (("", logits), ("_autocast", logits_amp)):
            leaf = lg.clone().requires_grad_(True)
            loss = fn(leaf, labels)
            loss.backward()
            blob[f"loss_{kind}{tag}"] = np.float64(loss.item())
            blob[f"dlogits_{kind}{tag}"] = leaf.grad.numpy()
            gn = grads_per_sample(model, batch, leaf.grad, autocast=bool(tag))
            blob[f"grad_norm_names_{kind}{tag}"] = np.array(list(gn.keys()))
            blob[f"grad_norm_values_{kind}{tag}"] = np.array(list(gn.values()), dtype=np.float64)
            print(f"  {name}: {kind}{tag} loss={loss.item():.6f} ({time.time() - t0:.0f}s)", flush=True)
    out = os.path.join(HERE, {"cfg2": "full_distilbert_cfg2.npz", "cfg3": "full_distilbert_cfg3.npz", "cfg4": "full_bert_cfg4.npz"}[name])
    np.savez_compressed(out, **blob)
    print(f"  wrote {out} ({os.path.getsize(out)} bytes)", flush=True)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_losses, NwayDualEncoder = MG.import_reference()
    for name in (sys.argv[1:] or list(CONFIGS)):
        make(name, NwayDualEncoder, ref_losses)


if __name__ == "__main__":
    main()
