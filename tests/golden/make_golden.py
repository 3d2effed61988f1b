#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by IMPORTING THE REFERENCE (build container only).

Usage (from the repo root, in the build container where /root/reference exists):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--skip-full]

What runs here is the real reference code: ``/root/reference/losses/*.py`` and
``/root/reference/models/nway_dual_encoder.py`` (which instantiates the HuggingFace encoder through
``AutoModel.from_pretrained``), driven with inputs from the portable generator in
``cl-drd_amd/synthetic.py``.  Only inputs and outputs are written (``*.npz``); no reference source or
bytecode leaves the container.  The one shim needed: ``transformers.AdamW`` (imported but unused by
``models/nway_dual_encoder.py:4``) was removed in transformers 5.x, so it is aliased to
``torch.optim.AdamW`` before the import.
"""
import argparse
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import cldrd_amd.synthetic as syn  # noqa: E402


def import_reference():
    import transformers
    sys.modules["transformers"].AdamW = torch.optim.AdamW
    sys.path.insert(0, REF)
    import losses as ref_losses
    from models.nway_dual_encoder import NwayDualEncoder
    return ref_losses, NwayDualEncoder


def t(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32)


def value_and_grad(fn, y_pred, *rest, **kw):
    yp = t(y_pred).clone().requires_grad_(True)
    out = fn(yp, *rest, **kw)
    out.backward()
    return np.float64(out.item()), yp.grad.numpy().astype(np.float32)


def make_losses(ref_losses, out):
    cases = {}
    # --- the reference's own __main__ demo inputs (known answers: SURVEY.md section 8c) ---
    M_s = [[2.0, 1.0, 1.0], [3.0, 1.5, 2.5]]
    M_t = [[2.5, 1.5, 2.0], [3.0, 2.0, 2.5]]
    cases["demo_kl"] = ("kl", np.array(M_s, np.float32), np.array(M_t, np.float32), dict(T=1.0))
    cases["demo_mse"] = ("mse", np.array(M_s, np.float32), np.array(M_t, np.float32), {})
    rn_pred = np.array([[103.8560, 104.2479, 102.9454, 103.0578, 98.6101, 100.2017, 100.1513, 100.0354, 99.1560,
                         101.1047, 97.7531, 98.9953, 101.6970, 101.1184, 98.9523, 98.2248, 99.3415, 98.2269, 98.9324,
                         97.9243, 99.5813, 95.6870, 99.5487, 101.5185, 96.9145, 102.6490, 100.5021, 97.7515, 97.8676,
                         99.5976],
                        [105.8982, 105.9335, 105.2820, 106.2369, 103.3414, 105.1359, 105.7083, 103.9510, 105.5665,
                         105.3788, 104.6647, 104.4636, 102.8736, 104.4074, 103.8423, 104.3142, 104.2956, 102.9430,
                         103.5177, 105.1869, 105.0547, 104.9325, 104.3588, 104.5267, 104.2974, 103.2128, 102.7218,
                         104.0699, 103.0756, 105.6170]], np.float32)
    rn_true = np.array([[6.2734, 6.2188, 6.0039, 4.9336, 3.6836, 3.3691, 3.3047, 3.2852, 3.2480, 3.0371, 2.5020,
                         2.1699, 2.0488, 1.9375, 1.9375, 1.7100, 1.5947, 1.5781, 1.5205, 1.4004, 1.3730, 1.3105,
                         1.3027, 1.2744, 1.2715, 1.2705, 1.0928, 1.0557, 0.9521, 0.9409],
                        [8.2500, 8.2188, 8.0703, 7.9375, 7.8906, 7.7969, 7.7344, 7.7070, 7.6562, 7.6484, 7.4609,
                         7.4102, 7.3789, 7.2930, 7.2383, 7.2148, 7.1836, 7.1836, 7.0391, 6.9570, 6.9453, 6.9414,
                         6.7930, 6.7539, 6.6797, 6.6367, 6.5547, 6.5430, 6.4531, 6.3438]], np.float32)
    cases["demo_ranknet"] = ("ranknet", rn_pred, rn_true, {})
    lt = np.array([[1., 0.5, 0., 0.], [1., 0.5, 0., 0.]], np.float32)
    p3 = np.array([[2.01, 1.23, 1.02, 0.4], [0.45, 1.04, 1.02, 3.12]], np.float32)
    p4 = np.array([[1.23, 2.01, 0.4, 1.02], [3.12, 1.04, 1.02, 0.45]], np.float32)
    bw = np.array([0.9, 1.3], np.float32)
    cases["demo_bweight3"] = ("bweight", p3, lt, dict(batch_weight=bw))
    cases["demo_bweight4"] = ("bweight", p4, lt, dict(batch_weight=bw))
    cases["demo_lambda3"] = ("lambda", p3, lt, {})
    # --- seeded cases at the BASELINE shapes ---
    for (B, N) in [(4, 8), (8, 32), (4, 64), (4, 200), (3, 30), (1, 5)]:
        seed = 1000 * B + N
        pred = (syn.normal(seed, B * N).reshape(B, N) * 3.0 + 100.0).astype(np.float32)
        teach = syn.teacher_scores(seed + 7, B, N)
        lab = syn.labels_mode9(B, N)
        for T in (1.0, 50.0):
            cases[f"kl_{B}x{N}_T{int(T)}"] = ("kl", pred, teach, dict(T=T))
        cases[f"mse_{B}x{N}"] = ("mse", pred, teach, {})
        cases[f"ranknet_{B}x{N}"] = ("ranknet", pred, teach, {})
        cases[f"ranknet_sum_{B}x{N}"] = ("ranknet", pred, teach, dict(reduction="sum"))
        cases[f"ranknet_lab_{B}x{N}"] = ("ranknet", pred, lab, {})
        cases[f"lambda_{B}x{N}"] = ("lambda", pred, lab, {})
        cases[f"lambda_sum_{B}x{N}"] = ("lambda", pred, teach, dict(reduction="sum"))
        w = (0.5 + syn.uniform01(seed + 9, B)).astype(np.float32)
        cases[f"bweight_{B}x{N}"] = ("bweight", pred, lab, dict(batch_weight=w))
        # padded (-1) entries: only lambda_mrr accepts them (losses/lambda_rank.py:66)
        labp = lab.copy()
        labp[:, -max(1, N // 4):] = -1.0
        labp[0, 1] = -1.0
        cases[f"lambda_pad_{B}x{N}"] = ("lambda", pred, labp, {})
        # in-batch negatives label fill of -0.5 (trainer :341-344)
        cases[f"lambda_inbatch_{B}x{N}"] = ("lambda", np.concatenate([pred, pred[:, ::-1] - 1.5], 1),
                                            np.concatenate([lab, np.full((B, N), -0.5, np.float32)], 1), {})
    fns = {"kl": lambda yp, yt, T=1.0: ref_losses.KLDiv(T)(yp, t(yt)),
           "mse": lambda yp, yt: ref_losses.MarginMSE()(yp, t(yt)),
           "ranknet": lambda yp, yt, **kw: ref_losses.ranknet_loss(yp, t(yt), **kw),
           "lambda": lambda yp, yt, **kw: ref_losses.lambda_mrr_loss(yp, t(yt), **kw),
           "bweight": lambda yp, yt, batch_weight=None, **kw: ref_losses.bweight_lambda_mrr_loss(yp, t(yt), t(batch_weight), **kw)}
    blob = {}
    names = []
    for name, (kind, pred, true, kw) in cases.items():
        val, grad = value_and_grad(fns[kind], pred, true, **kw)
        names.append(name)
        blob[name + "/kind"] = np.array(kind)
        blob[name + "/y_pred"] = pred
        blob[name + "/y_true"] = true
        for k, v in kw.items():
            blob[name + "/kw_" + k] = np.array(v)
        blob[name + "/value"] = val
        blob[name + "/grad"] = grad
        print(f"  {name:28s} value={val:.8f}")
    blob["names"] = np.array(names)
    np.savez_compressed(out, **blob)


def hf_model(arch, cfgd, seed):
    """HF model with weights from the portable generator; returns (model, state_dict names)."""
    from transformers import BertConfig, BertModel, DistilBertConfig, DistilBertModel
    if arch == "distilbert":
        cfg = DistilBertConfig(vocab_size=cfgd["vocab_size"], dim=cfgd["dim"], n_heads=cfgd["n_heads"],
                               hidden_dim=cfgd["hidden_dim"], n_layers=cfgd["n_layers"],
                               max_position_embeddings=cfgd["max_position_embeddings"],
                               dropout=0.0, attention_dropout=0.0)
        model = DistilBertModel(cfg)
    else:
        cfg = BertConfig(vocab_size=cfgd["vocab_size"], hidden_size=cfgd["dim"], num_attention_heads=cfgd["n_heads"],
                         intermediate_size=cfgd["hidden_dim"], num_hidden_layers=cfgd["n_layers"],
                         max_position_embeddings=cfgd["max_position_embeddings"],
                         hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
        model = BertModel(cfg)
    sd = model.state_dict()
    new = {}
    for k, v in sd.items():
        if not v.dtype.is_floating_point:
            new[k] = v
            continue
        new[k] = syn.init_param(seed, k, tuple(v.shape), std=cfgd.get("std", 0.02), perturb=True)
    model.load_state_dict(new)
    return model


def make_model_golden(NwayDualEncoder, ref_losses, out, arch, cfgd, B, N, Lq, Lp, ragged, store_grads, loss_kind="mse"):
    with tempfile.TemporaryDirectory() as tmp:
        qdir, pdir = os.path.join(tmp, "q"), os.path.join(tmp, "p")
        hf_model(arch, cfgd, seed=11).save_pretrained(qdir)
        hf_model(arch, cfgd, seed=12).save_pretrained(pdir)
        model = NwayDualEncoder(qdir, share_weights=False)
        # the reference loads both towers from ONE path; give the passage tower its own seeded weights
        from transformers import AutoModel
        model.passage_encoder = AutoModel.from_pretrained(pdir)
        model.eval()   # dropout off: parity is defined with dropout = 0
        batch = syn.nway_batch(4680, B, N, Lq, Lp, vocab=cfgd["vocab_size"], ragged=ragged)
        blob = {"arch": np.array(arch)}
        for k, v in cfgd.items():
            blob["cfg/" + k] = np.array(v)
        blob.update({"B": B, "N": N, "Lq": Lq, "Lp": Lp, "ragged": ragged, "loss_kind": np.array(loss_kind)})
        logits = model(batch["query"], batch["nway_passages"])
        if loss_kind == "mse":
            loss = ref_losses.MarginMSE()(logits, batch["labels"])
        elif loss_kind == "kl":
            loss = ref_losses.KLDiv(1.0)(logits, batch["labels"])
        else:
            loss = ref_losses.lambda_mrr_loss(logits, batch["labels"])
        model.zero_grad()
        loss.backward()
        blob["logits"] = logits.detach().numpy()
        blob["loss"] = np.float64(loss.item())
        # the reference's own mixed-precision path (it trains under autocast, nway_listwise_1.py:334); CPU autocast
        # only offers bf16, which is also the build's compute type: this is the drift the reference itself accepts
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
            blob["logits_autocast_bf16"] = model(batch["query"], batch["nway_passages"]).float().numpy()
        with torch.no_grad():
            blob["q_cls"] = model.query_embs(batch["query"]).numpy()
            blob["p_cls"] = model.nway_passage_embs(batch["nway_passages"]).numpy()
            if B > 1:
                for all_neg in (True, False):
                    m2 = NwayDualEncoder.__new__(NwayDualEncoder)
                    torch.nn.Module.__init__(m2)
                    m2.in_batch_loss, m2.all_in_batch_neg = True, all_neg
                    m2.query_encoder, m2.passage_encoder = model.query_encoder, model.passage_encoder
                    blob["logits_inbatch_all" if all_neg else "logits_inbatch_next"] = \
                        m2(batch["query"], batch["nway_passages"]).numpy()
        gn = {}
        for tower, enc in (("query_encoder", model.query_encoder), ("passage_encoder", model.passage_encoder)):
            for k, p in enc.named_parameters():
                if p.grad is None:
                    continue   # BERT pooler: no gradient on this path
                gn[f"{tower}.{k}"] = float(p.grad.norm())
                if store_grads:
                    blob[f"grad/{tower}.{k}"] = p.grad.numpy()
        blob["grad_norm_names"] = np.array(list(gn.keys()))
        blob["grad_norm_values"] = np.array(list(gn.values()), dtype=np.float64)
        np.savez_compressed(out, **blob)
        print(f"  {os.path.basename(out)}: loss={loss.item():.6f} logits[0,:3]={logits[0, :3].tolist()}")


def make_lr(out):
    from transformers import get_linear_schedule_with_warmup
    p = torch.nn.Parameter(torch.zeros(1))
    blob = {}
    for name, (warm, total) in {"a": (4000, 100000), "b": (10, 50), "c": (0, 7)}.items():
        opt = torch.optim.SGD([p], lr=1.0)
        sch = get_linear_schedule_with_warmup(opt, num_warmup_steps=warm, num_training_steps=total)
        steps = sorted(set([0, 1, 2, warm - 1, warm, warm + 1, total // 2, total - 1, total, total + 5]) & set(range(0, total + 6)))
        vals = []
        cur = 0
        for s in range(0, max(steps) + 1):
            if s in steps:
                vals.append(sch.get_last_lr()[0])
            opt.step()
            sch.step()
        blob[name + "/warmup"], blob[name + "/total"] = warm, total
        blob[name + "/steps"] = np.array(steps)
        blob[name + "/factor"] = np.array(vals, dtype=np.float64)
    np.savez_compressed(out, **blob)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-full", action="store_true")
    args = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_losses, NwayDualEncoder = import_reference()
    print("losses:")
    make_losses(ref_losses, os.path.join(HERE, "losses.npz"))
    print("lr schedule:")
    make_lr(os.path.join(HERE, "lr_schedule.npz"))
    print("models:")
    tiny = dict(vocab_size=128, dim=64, n_heads=2, hidden_dim=128, n_layers=2, max_position_embeddings=32, std=0.2)
    make_model_golden(NwayDualEncoder, ref_losses, os.path.join(HERE, "tiny_distilbert.npz"), "distilbert", tiny,
                      B=3, N=4, Lq=8, Lp=16, ragged=True, store_grads=True, loss_kind="mse")
    make_model_golden(NwayDualEncoder, ref_losses, os.path.join(HERE, "tiny_bert.npz"), "bert", tiny,
                      B=2, N=3, Lq=8, Lp=16, ragged=True, store_grads=True, loss_kind="lambda")
    if not args.skip_full:
        full = dict(vocab_size=30522, dim=768, n_heads=12, hidden_dim=3072, n_layers=6, max_position_embeddings=512)
        # cfg1 of BASELINE.json: B=4, N=8, L=128, MarginMSE, fp32 CPU
        make_model_golden(NwayDualEncoder, ref_losses, os.path.join(HERE, "full_distilbert_cfg1.npz"), "distilbert", full,
                          B=4, N=8, Lq=30, Lp=128, ragged=True, store_grads=False, loss_kind="mse")


if __name__ == "__main__":
    main()
