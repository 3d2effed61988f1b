#!/usr/bin/env python3
"""What parts of the training step cost on the critical path: the cfg2 step (graph replay) with parts SKIPPED (timing only, wrong
results): the query tower's forward / backward (side stream), the AdamW + norm tail.  One process, interleaved rounds.

Modes: full (the default step: the query tower runs free on the second stream) | win<a><l> (round-6 experiment: the query tower in slices of
a / l groups released at the passage tower's attention / LayerNorm launches, e.g. win32, win21, win43, win77) | no_q_bwd | no_q | no_opt
(parts skipped) | opt_side (AdamW moved to the head of the next step on
the second stream: timing only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cldrd_amd.synthetic as syn
from cldrd_amd.encoder import EncoderConfig
from cldrd_amd.models import NwayDualEncoder
from cldrd_amd.trainer import NwayTrainer

dev = torch.device("cuda")
B, N, L, Lq = 8, 32, 128, 30


def build(mode):
    torch.manual_seed(0)
    model = NwayDualEncoder(EncoderConfig(arch="distilbert"), share_weights=False).to(dev).train()
    tr = NwayTrainer(model, loss="kl_div", T=1.0, learning_rate=7e-6, warmup_steps=4000, total_steps=100000)
    qe = model.query_encoder
    if mode.startswith("qprio"):
        # the second (query tower) stream at another HIP stream priority: qprio-1 high, qprio1 low (round 3 tried high: no change)
        tr.q_stream = torch.cuda.Stream(device=dev, priority=int(mode[5:]))
        tr.flat_g.record_stream(tr.q_stream)
    if mode.startswith("win") and len(mode) == 5:
        tr.window_schedule = True
        tr.FWD_SLICE = tr.BWD_SLICE = {"attn": int(mode[3]), "ln": int(mode[4])}
    if mode in ("no_q", "no_q_bwd"):
        real_enc, real_bwd = qe.encode, qe.backward_from_cls
        cache = {}
        if mode == "no_q":
            def enc(*a, **k):
                if "out" not in cache:
                    cache["out"] = real_enc(*a, **k)
                return cache["out"]
            qe.encode = enc
        qe.backward_from_cls = lambda *a, **k: None
    if mode == "no_opt":
        tr._optimizer_launches = lambda *a, **k: None
    if mode == "opt_side":
        # VERDICT r05 item 1c, as a timing experiment (WRONG results: the forward reads weights while they are updated): AdamW of step t - 1 at the
        # HEAD of step t on the second stream, free to overlap the whole forward (no per-layer waits: the most optimistic schedule); the clip
        # norm stays at the tail, where it has to be.  What this mode gains over `full` is the ceiling of what a deferred optimizer can return.
        from cldrd_amd import hip_ops as ops
        real_fb = tr.forward_backward

        def fb(b, **k):
            side = tr.q_stream
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                tr._adamw_launches(1e-6, 10, tr.model.towers(), tr._state["hyper"] if tr._state else None)
            return real_fb(b, **k)

        def tail(lr, step):
            with ops.loss_scale(tr._scale_state.data_ptr() if tr.amp16 else None, tr.scale_growth_interval):
                tr._norm_launches()
        tr.forward_backward, tr._optimizer_launches = fb, tail
    return tr


batch = syn.nway_batch(4680, B, N, Lq, L, ragged=False, label_kind="teacher")
batch = {k: ({kk: vv.to(dev) for kk, vv in v.items()} if isinstance(v, dict) else v.to(dev)) for k, v in batch.items()}
modes = [a for a in sys.argv[1:] if not a.isdigit()] or ["full", "no_q_bwd", "no_q", "no_opt"]
ROUNDS = next((int(a) for a in sys.argv[1:] if a.isdigit()), 3)
trs = {m: build(m) for m in modes}
for m, tr in trs.items():
    for _ in range(8):
        tr.train_step(batch)
torch.cuda.synchronize()
res = {m: [] for m in modes}
for rnd in range(ROUNDS):
    for m, tr in trs.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            tr.train_step(batch)
        torch.cuda.synchronize()
        res[m].append((time.perf_counter() - t0) / 40 * 1e3)
for m in modes:
    print(f"{m:10s} ms/step min {min(res[m]):.3f}  all {[round(x, 3) for x in res[m]]}")
