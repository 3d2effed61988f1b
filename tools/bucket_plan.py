#!/usr/bin/env python3
"""Bucket plan of the data-parallel gradient all-reduce (reference: DDP wrap, trainer/multistep-curriculum/nway_listwise_1.py:250-255) for
cfg2 / cfg3 / cfg4, printed WITHOUT a GPU: bucket = one contiguous slice of the joint flat gradient buffer (one per transformer layer and
tower, plus each tower's embedding block), in the order the backward completes them, with the backward kernels each all-reduce is issued
behind (= what it can overlap with) and the bytes a ring / a direct exchange would move per rank over xGMI.  A baseline for the 8-GPU run
the pool cannot give: compare the launch order and bytes an rccl trace shows with this table.

usage: python tools/bucket_plan.py [cfg2 cfg3 cfg4] [--world 8]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cldrd_amd.encoder import EncoderConfig, FlatLayout       # noqa: E402  (host-side layout only: no GPU, no library)

CONFIGS = {
    "cfg2": dict(arch="distilbert", n_layers=6, B=8, N=32, L=128, Lq=30, loss="kl_div"),
    "cfg3": dict(arch="distilbert", n_layers=6, B=4, N=200, L=128, Lq=30, loss="margin_mse"),
    "cfg4": dict(arch="bert", n_layers=12, B=4, N=64, L=256, Lq=30, loss="ranknet / lambda_mrr"),
}
XGMI_LINK_GBS = 153.0          # per link and direction (MI355X_MICROARCH / task notes: 7 links x ~153 GB/s per GPU)


def plan(name, world):
    c = CONFIGS[name]
    cfg = EncoderConfig(arch=c["arch"], n_layers=c["n_layers"])
    lay = FlatLayout(cfg)
    flush = max(1, -(-cfg.n_layers // 2))          # NwayTrainer: weight gradients flushed every ceil(layers / 2) layers on data-parallel ranks
    towers = (("query", c["B"] * c["Lq"]), ("passage", c["B"] * c["N"] * c["L"]))
    rows = []
    order = 0
    # launch order inside one step (trainer._backward, eager ranks: the query tower's backward is enqueued first on the second stream, the
    # passage tower's on the main stream; each tower: layers n-1 .. 0 complete in groups of `flush`, then the embedding block)
    for tname, tokens in towers:
        toff = 0 if tname == "query" else lay.total
        waiting = []

        def emit(group):
            nonlocal order
            behind = f"weight-gradient group of layers {group[0]}..{group[-1]} ({4 * len(group)}+ problems, T = {tokens} tokens)"
            for j in group:
                a, b = lay.layer_range[j]
                rows.append((order, tname, f"layer {j}", toff + a, toff + b, behind))
                order += 1
        # mirrors HipEncoder.backward_from_cls: a layer's hook runs once its weight gradients have been LAUNCHED (every `flush` layers); the
        # embedding block's hook runs right behind embed_ln_bwd, in front of whatever weight-gradient group is still waiting
        for li in reversed(range(cfg.n_layers)):
            waiting.append(li)
            if len(waiting) >= flush and li != 0:      # layer 0 never triggers an early flush (encoder.py: layer_done)
                emit(waiting)
                waiting = []
        a, b = lay.embed_range
        rows.append((order, tname, "embeddings", toff + a, toff + b, "embed_ln_bwd" + (" (in front of the tower's last weight-gradient group)" if waiting else "")))
        order += 1
        if waiting:
            emit(waiting)
    total = sum(r[4] - r[3] for r in rows) * 4
    print(f"== {name}: {c['arch']} x2 towers (unshared), {cfg.n_layers} layers, per-GPU batch {c['B']}, N = {c['N']}, L = {c['L']}, loss {c['loss']}; world {world}")
    print(f"   gradient bytes per step {total / 1e6:.1f} MB fp32 in {len(rows)} buckets; 1 / world folded into dlogits (SUM all-reduce)")
    print(f"   ring all-reduce: 2 (w-1)/w S = {2 * (world - 1) / world * total / 1e6:.1f} MB sent per rank over ONE link pair -> "
          f">= {2 * (world - 1) / world * total / XGMI_LINK_GBS / 1e6:.2f} ms; direct reduce-scatter + all-gather over all {world - 1} links: "
          f">= {2 * (world - 1) / world * total / (XGMI_LINK_GBS * (world - 1)) / 1e6:.2f} ms (link-bound floors, no latency)")
    print(f"   {'#':>3s} {'tower':8s} {'bucket':11s} {'offset':>12s} {'MB':>8s}  issued behind (what the collective overlaps: everything enqueued after it)")
    for o, t, bname, a, b, behind in rows:
        print(f"   {o:3d} {t:8s} {bname:11s} {a:12d} {(b - a) * 4 / 1e6:8.2f}  {behind}")
    print("   then: Work.wait() of every bucket on the main stream -> clip norm -> AdamW (trainer._wait_pending)")
    print()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", default=["cfg2", "cfg3", "cfg4"])
    ap.add_argument("--world", type=int, default=8)
    a = ap.parse_args()
    for n in a.configs:
        plan(n, a.world)


if __name__ == "__main__":
    main()
