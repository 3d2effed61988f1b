#!/usr/bin/env python3
"""Layer-by-layer relative error of the CLS row (and of all valid rows) of one tower: GPU vs fp32 oracle, next to the CPU emulation
of the modelled rounding points.  Finds the kernel that adds error beyond the bf16-operand floor."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import cldrd_amd.synthetic as syn  # noqa: E402
import drift_emulation as DE  # noqa: E402
from cldrd_amd.encoder import EncoderConfig, HipEncoder  # noqa: E402
from oracle import encoder_ref as E  # noqa: E402

torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))


def rel(a, b):
    return float(np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2)))


for arch, layers, M, L, ragged in (("bert", 6, 8, 30, True), ("distilbert", 6, 8, 30, True), ("bert", 6, 8, 30, False), ("bert", 6, 8, 128, True)):
    rcfg = E.RefConfig(arch=arch, n_layers=layers)
    w = {k: syn.init_param(11, k, s, std=0.02, perturb=True) for k, s in E.param_shapes(rcfg).items()}
    b = syn.nway_batch(4680, M, 2, L, L, ragged=ragged)
    ids, mask = b["query"]["input_ids"], b["query"]["attention_mask"]
    if L > 30:
        ids, mask = b["nway_passages"]["input_ids"][:, 0], b["nway_passages"]["attention_mask"][:, 0]
    with torch.no_grad():
        _, ref_h = E.encoder_forward(w, rcfg, ids, mask, return_all=True)
        emu_h = []
        emu_cls = DE.forward(w, rcfg, ids, mask, dict(sum32=True, res32=True), hiddens=emu_h)
    enc = HipEncoder(EncoderConfig(arch=arch, n_layers=layers, dropout=0.0, attention_dropout=0.0))
    with torch.no_grad():
        for name, prm in enc.named_flat():
            prm.copy_(syn.init_param(11, name, tuple(prm.shape), std=0.02, perturb=True))
    enc.cuda().eval()
    print(f"== {arch} {layers} layers, M {M}, L {L}, ragged {int(ragged)} (valid tokens per row: {mask.sum(1).tolist()})")
    for cls_only in (False, True):
        enc.cls_only_last = cls_only
        with torch.no_grad():
            cls, tape = enc.encode(ids.cuda(), mask.cuda(), train=False, save=True)
        T = M * L
        valid = mask.reshape(-1).numpy().astype(bool)
        for i, a in enumerate(tape.layers):
            x = a["x_in"][:T].float().cpu().numpy()
            r, e = ref_h[i].reshape(T, -1).numpy(), emu_h[i].reshape(T, -1).numpy()
            crow = np.arange(M) * L
            print(f"   cls_only {int(cls_only)} layer {i} input: CLS rows GPU {rel(x[crow], r[crow]):.2e} emu {rel(e[crow], r[crow]):.2e} | "
                  f"valid rows GPU {rel(x[valid], r[valid]):.2e} emu {rel(e[valid], r[valid]):.2e}")
        rc = ref_h[-1][:, 0].numpy()
        print(f"   cls_only {int(cls_only)} output CLS: GPU {rel(cls.cpu().numpy(), rc):.2e} emu {rel(emu_cls.numpy(), rc):.2e}", flush=True)
