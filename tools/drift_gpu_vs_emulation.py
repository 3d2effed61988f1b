#!/usr/bin/env python3
"""Where does the GPU's logit error come from?  For a few (arch, layers, L) variants: fp32 oracle logits (CPU), the CPU emulation
of the pipeline's rounding points (tools/drift_emulation.py: the error floor of bf16-operand GEMMs with an fp32 residual stream),
and the GPU logits.  GPU rms error / emulation rms error ~ 1 means the kernels add nothing beyond the modelled roundings."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import cldrd_amd.synthetic as syn  # noqa: E402
import drift_emulation as DE  # noqa: E402
from cldrd_amd.encoder import EncoderConfig  # noqa: E402
from cldrd_amd.models import NwayDualEncoder  # noqa: E402
from oracle import encoder_ref as E  # noqa: E402

torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
B, N, Lq = 2, 16, 30
for arch, layers, Lp, ragged in (("distilbert", 6, 128, True), ("distilbert", 6, 256, True), ("bert", 6, 128, True), ("bert", 12, 256, True),
                                 ("distilbert", 6, 256, False), ("distilbert", 1, 256, True), ("distilbert", 1, 128, True)):
    rcfg = E.RefConfig(arch=arch, n_layers=layers)
    shapes = E.param_shapes(rcfg)
    qp = {k: syn.init_param(11, k, s, std=0.02, perturb=True) for k, s in shapes.items()}
    pp = {k: syn.init_param(12, k, s, std=0.02, perturb=True) for k, s in shapes.items()}
    batch = syn.nway_batch(4680, B, N, Lq, Lp, ragged=ragged)
    with torch.no_grad():
        ref = E.nway_forward(qp, pp, rcfg, batch["query"], batch["nway_passages"]).numpy()
        mode = dict(sum32=True, res32=True)
        q = DE.forward(qp, rcfg, batch["query"]["input_ids"], batch["query"]["attention_mask"], mode)
        p = DE.forward(pp, rcfg, batch["nway_passages"]["input_ids"].reshape(B * N, Lp),
                       batch["nway_passages"]["attention_mask"].reshape(B * N, Lp), mode).view(B, N, -1)
        emu = torch.sum(q.unsqueeze(1) * p, dim=-1).numpy()
        pcls_ref = E.nway_passage_embs(pp, rcfg, batch["nway_passages"]).numpy()
    cfg = EncoderConfig(arch=arch, n_layers=layers, dropout=0.0, attention_dropout=0.0)
    model = NwayDualEncoder(cfg, share_weights=False)
    with torch.no_grad():
        for seed, tower in ((11, model.query_encoder), (12, model.passage_encoder)):
            for name, prm in tower.named_flat():
                prm.copy_(syn.init_param(seed, name, tuple(prm.shape), std=0.02, perturb=True))
    model.cuda().eval()
    dev_batch = {k: ({kk: vv.cuda() for kk, vv in v.items()} if isinstance(v, dict) else v.cuda()) for k, v in batch.items()}
    outs = {}
    for cls_only in (True, False):
        for t in model.towers():
            t.cls_only_last = cls_only
        with torch.no_grad():
            outs[cls_only] = model(dev_batch["query"], dev_batch["nway_passages"]).cpu().numpy()
            pc = model.nway_passage_embs(dev_batch["nway_passages"]).cpu().numpy()
        if cls_only:
            pcls_err = np.sqrt(np.mean((pc - pcls_ref) ** 2)) / np.sqrt(np.mean(pcls_ref ** 2))
            pcls_emu = np.sqrt(np.mean((p.numpy() - pcls_ref) ** 2)) / np.sqrt(np.mean(pcls_ref ** 2))
    rms = lambda a: float(np.sqrt(np.mean((a - ref) ** 2)))
    print(f"{arch:10s} layers {layers:2d} L {Lp} ragged {int(ragged)}: rms error emulation {rms(emu):.4f} | GPU cls-only {rms(outs[True]):.4f} "
          f"({rms(outs[True]) / rms(emu):.2f}x) | GPU full last layer {rms(outs[False]):.4f} ({rms(outs[False]) / rms(emu):.2f}x) | "
          f"passage CLS rel rms: GPU {pcls_err:.2e} emulation {pcls_emu:.2e}", flush=True)
