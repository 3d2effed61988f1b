import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_model as T
from cldrd_amd.trainer import NwayTrainer
import cldrd_amd.synthetic as syn
for name in [a for a in sys.argv[1:] if not a.startswith('-')]:
    fname, arch, layers, kinds = T.FULL_CONFIGS[name]
    g = np.load(os.path.join(T.GOLDEN, fname))
    model = T._full_size_model(arch, layers)
    B, N, Lq, Lp = int(g["B"]), int(g["N"]), int(g["Lq"]), int(g["Lp"])
    label_kind = str(g["label_kind"]) if "label_kind" in g.files else "teacher"
    batch = syn.nway_batch(4680, B, N, Lq, Lp, ragged=True, label_kind=label_kind)
    for gk, loss_kind in kinds:
        if f"gslice_names_{gk}" not in g.files: continue
        tr = NwayTrainer(model, loss=loss_kind)
        tr.forward_backward(batch)
        params = {f"query_encoder.{n}": p for n, p in model.query_encoder.named_flat()}
        params.update({f"passage_encoder.{n}": p for n, p in model.passage_encoder.named_flat()})
        rows = []
        for n in [str(x) for x in g[f"gslice_names_{gk}"]]:
            want = g[f"gslice/{gk}/{n}"].astype(np.float64)
            have = params[n].grad.detach()
            have = (have if have.dim() == 1 else have[:want.shape[0]]).double().cpu().numpy()
            c = float((want * have).sum() / (np.linalg.norm(want) * np.linalg.norm(have) + 1e-300))
            a = g[f"gslice_autocast/{gk}/{n}"].astype(np.float64)
            ca = float((want * a).sum() / (np.linalg.norm(want) * np.linalg.norm(a) + 1e-300))
            rows.append((n, c, ca, float(np.linalg.norm(want))))
            if c < 0.9995 and "-v" in os.environ.get("COS_FLAGS", ""):
                print(f"{name}/{loss_kind} {n:70s} cos {c:.5f} (ref bf16 autocast {ca:.5f}) |g| {np.linalg.norm(want):.3e}")
        # tensors whose exact gradient is zero up to rounding (k_lin.bias: softmax is shift-invariant; the last layer's output LayerNorm bias of
        # a CLS-only loss outside the CLS row) are noise in every implementation: listed, not ranked
        gmax = max(r[3] for r in rows)
        live = [r for r in rows if r[3] > 1e-4 * gmax]
        dead = [r for r in rows if r[3] <= 1e-4 * gmax]
        cs = np.array([r[1] for r in live]); ca_ = np.array([r[2] for r in live])
        worst = min(live, key=lambda r: r[1])
        print(f"{name}/{loss_kind} mode={os.environ.get('CLDRD_AMP', 'fp16(default)')}: {len(live)} tensors with |g| > 1e-4 max|g|: min cos {cs.min():.5f} ({worst[0]}), median {np.median(cs):.5f}; "
              f"reference under bf16 autocast on the same tensors: min {ca_.min():.5f}, median {np.median(ca_):.5f}; {len(dead)} tensors with ~zero exact gradient not ranked")
