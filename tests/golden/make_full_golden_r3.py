#!/usr/bin/env python3
"""Round-3 additions to the full-size goldens, produced by IMPORTING THE REFERENCE (build container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_full_golden_r3.py [cfg1] [cfg2] [cfg3] [cfg4]

For every config the existing ``full_*.npz`` keeps all of its entries and gains

  * ``logits_autocast_fp16``: the reference's forward under ``torch.autocast("cpu", dtype=torch.float16)`` - the mixed-precision
    mode the reference actually trains in (fp16 autocast, nway_listwise_1.py:334); the bf16-autocast logits stored earlier are the
    drift of the same code at this package's operand width;
  * (every config since round 4) ``gslice/<kind>/<tower>.<parameter>``: fp32 gradients of the reference for a handful of parameters -
    every 1-D parameter of the first / middle / last layer (DistilBERT 0, 2, 5; BERT-base 0, 5, 11) and of the embedding LayerNorm in
    full, the first 16 rows of the weight matrices of the first and last layer and of the position embeddings - so that the GPU test can check gradient DIRECTIONS (cosine), not only norms;
    ``gslice_autocast/...``: the same slices from the reference's bf16-autocast forward + backward (the yardstick).

Weights and inputs come from the portable generator (cl-drd_amd/synthetic.py), so only outputs are stored.
"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402
import make_full_golden as MF  # noqa: E402

syn = MG.syn

FILES = {"cfg1": "full_distilbert_cfg1.npz", "cfg2": "full_distilbert_cfg2.npz", "cfg3": "full_distilbert_cfg3.npz", "cfg4": "full_bert_cfg4.npz"}
CONFIGS = dict(MF.CONFIGS)
CONFIGS["cfg1"] = ("distilbert", 4, 8, 30, 128, "teacher", ["mse"])        # make_golden.py: default label kind of nway_batch
SLICE_ROWS = 16


def wanted(name: str, n_layers: int = 6) -> bool:
    """First / middle / last layer: DistilBERT 0, 2, 5 (``transformer.layer.i.``), BERT-base 0, 5, 11 (``encoder.layer.i.``)."""
    if name.startswith("embeddings.LayerNorm") or name == "embeddings.position_embeddings.weight":
        return True
    first, mid, last = 0, (2 if n_layers <= 6 else n_layers // 2 - 1), n_layers - 1
    for i in (first, mid, last):
        if name.startswith(f"transformer.layer.{i}.") or name.startswith(f"encoder.layer.{i}."):
            if name.endswith(".bias") or "layer_norm" in name or "LayerNorm" in name:
                return True
            if i in (first, last) and name.endswith(".weight"):
                return True
    return False


def make(name, NwayDualEncoder, ref_losses):
    arch, B, N, Lq, Lp, label_kind, kinds = CONFIGS[name]
    cfgd = MF.FULL[arch]
    path = os.path.join(HERE, FILES[name])
    old = dict(np.load(path))
    t0 = time.time()
    with tempfile.TemporaryDirectory() as tmp:
        qdir, pdir = os.path.join(tmp, "q"), os.path.join(tmp, "p")
        MG.hf_model(arch, cfgd, seed=11).save_pretrained(qdir)
        MG.hf_model(arch, cfgd, seed=12).save_pretrained(pdir)
        model = NwayDualEncoder(qdir, share_weights=False)
        from transformers import AutoModel
        model.passage_encoder = AutoModel.from_pretrained(pdir)
    model.eval()
    kw = {} if name == "cfg1" else dict(label_kind=label_kind)
    batch = syn.nway_batch(4680, B, N, Lq, Lp, vocab=cfgd["vocab_size"], ragged=True, **kw)
    with torch.no_grad():
        logits = model(batch["query"], batch["nway_passages"])
        assert np.array_equal(logits.numpy(), old["logits"]), "the regenerated fp32 logits differ from the stored golden"
        with torch.autocast("cpu", dtype=torch.float16):
            l16 = model(batch["query"], batch["nway_passages"]).float()
    old["logits_autocast_fp16"] = l16.numpy()
    d16, dbf = (l16 - logits).abs().max().item(), np.abs(old["logits_autocast_bf16"] - old["logits"]).max()
    print(f"  {name}: fp16-autocast drift max {d16:.5f} rms {(l16 - logits).pow(2).mean().sqrt().item():.5f} | bf16-autocast drift max {dbf:.5f} "
          f"| max|logit| {logits.abs().max().item():.3f} ({time.time() - t0:.0f}s)", flush=True)
    nl = cfgd["n_layers"]
    if True:        # round 4: every config (cfg3 and cfg4 had per-tensor norms only)
        for kind in kinds:
            leaf = logits.clone().requires_grad_(True)
            loss = MF.loss_fn(ref_losses, kind)(leaf, batch["labels"])
            loss.backward()
            key = "dlogits_" + kind
            if key in old:
                assert np.array_equal(leaf.grad.numpy(), old[key])
            else:
                old[key] = leaf.grad.numpy()
            MF.grads_per_sample(model, batch, leaf.grad, autocast=False)
            names = []
            for tower, enc in (("query_encoder", model.query_encoder), ("passage_encoder", model.passage_encoder)):
                for k, p in enc.named_parameters():
                    if p.grad is None or not wanted(k, nl):
                        continue
                    g = p.grad if p.grad.dim() == 1 else p.grad[:SLICE_ROWS]
                    old[f"gslice/{kind}/{tower}.{k}"] = g.numpy().astype(np.float32).copy()
                    names.append(f"{tower}.{k}")
            old[f"gslice_names_{kind}"] = np.array(names)
            print(f"  {name}: {kind} gradient slices of {len(names)} parameters ({time.time() - t0:.0f}s)", flush=True)
            # the same slices from the reference's own bf16-autocast forward + backward (its loss on its autocast logits): how far a
            # 16-bit-operand run of the REFERENCE turns these gradients - the yardstick for ours
            with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
                lamp = model(batch["query"], batch["nway_passages"]).float()
            assert np.array_equal(lamp.numpy(), old["logits_autocast_bf16"])
            leaf = lamp.clone().requires_grad_(True)
            MF.loss_fn(ref_losses, kind)(leaf, batch["labels"]).backward()
            MF.grads_per_sample(model, batch, leaf.grad, autocast=True)
            for tower, enc in (("query_encoder", model.query_encoder), ("passage_encoder", model.passage_encoder)):
                for k, p in enc.named_parameters():
                    if p.grad is None or not wanted(k, nl):
                        continue
                    g = p.grad if p.grad.dim() == 1 else p.grad[:SLICE_ROWS]
                    old[f"gslice_autocast/{kind}/{tower}.{k}"] = g.float().numpy().astype(np.float32).copy()
            print(f"  {name}: {kind} autocast gradient slices ({time.time() - t0:.0f}s)", flush=True)
    np.savez_compressed(path, **old)
    print(f"  wrote {path} ({os.path.getsize(path)} bytes)", flush=True)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_losses, NwayDualEncoder = MG.import_reference()
    for name in (sys.argv[1:] or list(CONFIGS)):
        make(name, NwayDualEncoder, ref_losses)


if __name__ == "__main__":
    main()
