"""End-to-end parity of the hot path on the MI355X (run with -m gpu): NwayDualEncoder forward / backward and the
fused trainer step against the CPU oracle (oracle/, itself pinned to the reference by tests/golden/*.npz), plus
the full-size golden of BASELINE.json configs[0].

Tolerances: GPU bf16 vs fp32 oracle -> logits <= 5e-3 of the logit scale at full size
(looser on the tiny models, whose weights are 10x larger than the HF init), loss <= 1e-2 rel, gradient cosine >= 0.999."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import cldrd_amd.synthetic as syn
from cldrd_amd import hip_ops as ops
import selftest
from cldrd_amd.encoder import EncoderConfig
from cldrd_amd.models import NwayDualEncoder
from cldrd_amd.trainer import NwayTrainer
from oracle import encoder_ref as E
from oracle import losses_ref as LR
from oracle import optim_ref as O

from conftest import GOLDEN


def cos(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))


def small_cfg(arch="distilbert", layers=2):
    return EncoderConfig(arch=arch, vocab_size=512, dim=128, n_heads=2, hidden_dim=256, n_layers=layers,
                         max_position_embeddings=64, dropout=0.0, attention_dropout=0.0)


def oracle_run(model, cfg, batch, loss_kind, in_batch=False, all_neg=True, T=1.0):
    qp, pp = selftest.oracle_params(model)
    for p in set(list(qp.values()) + list(pp.values())):
        p.requires_grad_(True)
    logits = E.nway_forward(qp, pp, selftest.oracle_cfg(cfg), batch["query"], batch["nway_passages"], in_batch, all_neg)
    labels = batch["labels"].numpy()
    if in_batch:
        labels = np.concatenate([labels, np.full((labels.shape[0], logits.shape[1] - labels.shape[1]), -0.5, np.float32)], 1)
    fn = {"margin_mse": LR.margin_mse, "kl_div": lambda a, b: LR.kl_div(a, b, T), "lambda_mrr": LR.lambda_mrr,
          "ranknet": LR.ranknet}[loss_kind]
    val, dl = fn(logits.detach().numpy(), labels)
    logits.backward(torch.from_numpy(dl).float())
    return logits.detach().numpy(), val, qp, pp


@pytest.mark.parametrize("arch,loss_kind,share", [("distilbert", "margin_mse", False), ("bert", "margin_mse", False),
                                                   ("distilbert", "kl_div", True), ("distilbert", "lambda_mrr", False)])
def test_small_model_forward_backward_vs_oracle(arch, loss_kind, share):
    cfg = small_cfg(arch)
    model = selftest.build_tiny_model(cfg, share_weights=share).cuda()
    model.train()
    label_kind = "teacher" if loss_kind in ("margin_mse", "kl_div") else "mode9"
    batch = syn.nway_batch(4680, 3, 5, 10, 40, vocab=cfg.vocab_size, ragged=True, label_kind=label_kind)
    ref_logits, ref_loss, qp, pp = oracle_run(model, cfg, batch, loss_kind)
    tr = NwayTrainer(model, loss=loss_kind, learning_rate=1e-3, warmup_steps=0, total_steps=100)
    loss_out, logits = tr.forward_backward(batch)
    torch.cuda.synchronize()
    got = logits.cpu().numpy()
    scale = np.abs(ref_logits).max()
    assert np.abs(got - ref_logits).max() <= 2e-2 * scale, f"logits off: {np.abs(got - ref_logits).max() / scale:.3e}"
    assert loss_out[0].item() == pytest.approx(ref_loss, rel=5e-2, abs=1e-3)
    towers = [("query_encoder", model.query_encoder, qp)]
    if not share:
        towers.append(("passage_encoder", model.passage_encoder, pp))
    for tname, tower, ref_params in towers:
        tower.ensure_grads()
        all_g, all_r = [], []
        for name, p in tower.named_flat():
            gg, rr = p.grad.detach().cpu().numpy(), ref_params[name].grad.numpy()
            all_g.append(gg.ravel()), all_r.append(rr.ravel())
            if np.linalg.norm(rr) > 1e-3 * max(1.0, ref_loss):          # skip analytically-zero grads (k bias)
                # the token-type row is the sum of ALL token gradients (heavy cancellation): bf16 noise shows most there
                thr = 0.99 if "token_type" in name else 0.995
                if loss_kind == "lambda_mrr":
                    thr = 0.98      # rank weights are discontinuous in the logits: a bf16-level flip of two near-equal scores changes dlogits
                assert cos(gg, rr) > thr, f"{tname}.{name}: cosine {cos(gg, rr):.5f}"
                assert np.linalg.norm(gg) == pytest.approx(np.linalg.norm(rr), rel=5e-2), f"{tname}.{name}"
        assert cos(np.concatenate(all_g), np.concatenate(all_r)) > 0.999


@pytest.mark.parametrize("all_neg", [True, False])
def test_in_batch_negatives_vs_oracle(all_neg):
    cfg = small_cfg()
    model = selftest.build_tiny_model(cfg).cuda()
    model.in_batch_loss, model.all_in_batch_neg = True, all_neg
    model.train()
    batch = syn.nway_batch(4680, 3, 4, 8, 24, vocab=cfg.vocab_size, ragged=True, label_kind="mode9")
    ref_logits, ref_loss, qp, pp = oracle_run(model, cfg, batch, "lambda_mrr", True, all_neg)
    tr = NwayTrainer(model, loss="lambda_mrr")
    loss_out, logits = tr.forward_backward(batch)
    got = logits.cpu().numpy()
    assert got.shape == ref_logits.shape
    # tiny model with 10x the HF init scale: CLS embeddings carry ~0.9 % bf16 noise per element on both code paths
    # (full last layer and CLS-only), which the [B, B*N] dot products turn into up to ~2.5 % of the largest logit
    assert np.abs(got - ref_logits).max() <= 3e-2 * np.abs(ref_logits).max()
    g = np.concatenate([p.grad.cpu().numpy().ravel() for _, p in model.passage_encoder.named_flat()])
    r = np.concatenate([pp[n].grad.numpy().ravel() for n, _ in model.passage_encoder.named_flat()])
    assert cos(g, r) > 0.995


def test_autograd_bridge_matches_fused_path():
    """The reference-style loop (model(...) -> loss -> loss.backward()) and the fused trainer give the same gradients."""
    from cldrd_amd.losses import MarginMSE
    cfg = small_cfg()
    model = selftest.build_tiny_model(cfg).cuda()
    model.train()
    batch = syn.nway_batch(4680, 2, 3, 8, 32, vocab=cfg.vocab_size, ragged=True)
    dev_batch = {k: ({kk: vv.cuda() for kk, vv in v.items()} if isinstance(v, dict) else v.cuda()) for k, v in batch.items()}
    tr = NwayTrainer(model, loss="margin_mse")
    loss_out, logits = tr.forward_backward(batch)
    fused = tr.flat_g.clone()
    tr.flat_g.zero_()
    logits2 = model(dev_batch["query"], dev_batch["nway_passages"])
    loss = MarginMSE()(logits2, dev_batch["labels"])
    loss.backward()
    assert torch.allclose(logits2, logits, rtol=0, atol=0)
    assert loss.item() == pytest.approx(loss_out[0].item(), rel=1e-6)
    # embedding scatter uses float atomics (order-dependent in the last bits); everything else is bit-reproducible
    assert torch.allclose(tr.flat_g, fused, rtol=1e-4, atol=1e-6 * fused.abs().max().item())
    # eval-mode calls of the reference surface
    model.eval()
    with torch.no_grad():
        q = model.query_embs(dev_batch["query"])
        p = model.nway_passage_embs(dev_batch["nway_passages"])
        assert q.shape == (2, cfg.dim) and p.shape == (2, 3, cfg.dim) and q.dtype == torch.float32
        out = model.query_encoder(**dev_batch["query"])[0][:, 0, :]
        assert torch.equal(out, q)


def test_trainer_step_matches_oracle_update():
    """clip_grad_norm_ + legacy AdamW + linear schedule on the GPU vs the oracle restatement, from the GPU's own gradients."""
    cfg = small_cfg(layers=1)
    model = selftest.build_tiny_model(cfg).cuda()
    model.train()
    batch = syn.nway_batch(4680, 2, 3, 8, 16, vocab=cfg.vocab_size)
    tr = NwayTrainer(model, loss="margin_mse", learning_rate=1e-3, warmup_steps=2, total_steps=10, weight_decay=0.01)
    tr.global_step = tr.adam_step = 1          # lr factor 0.5 in effect
    p0 = tr.flat_p.clone()
    tr.forward_backward(batch)
    g = tr.flat_g.clone()
    lr = tr.optimizer_step()
    assert lr == pytest.approx(1e-3 * 0.5)
    total, coef = O.clip_coef([g.cpu().numpy()], 1.0)
    assert tr.clip[0].item() == pytest.approx(total, rel=1e-4)
    dec = (tr.decay_flags & 1).repeat_interleave(64).bool().cpu().numpy()        # bit 0: weight decay (bit 1: no 16-bit shadow)
    gn = g.double().cpu().numpy() * coef
    z = np.zeros_like(gn)
    p1, _, _ = O.adamw_step(p0.cpu().numpy(), gn, z, z, lr=lr, step=2, weight_decay=0.0)
    p2, _, _ = O.adamw_step(p0.cpu().numpy(), gn, z, z, lr=lr, step=2, weight_decay=0.01)
    ref = np.where(dec, p2, p1)
    assert np.allclose(tr.flat_p.cpu().numpy(), ref, rtol=2e-5, atol=2e-6)
    # decay flags follow the reference's name rule
    off, shape = model.query_encoder.layout.entries["transformer.layer.0.sa_layer_norm.weight"]
    assert dec[off]                       # decayed (does not match 'LayerNorm.weight')
    off, _ = model.query_encoder.layout.entries["embeddings.LayerNorm.weight"]
    assert not dec[off]
    off, _ = model.query_encoder.layout.entries["transformer.layer.0.ffn.lin1.bias"]
    assert not dec[off]
    # shadows follow the update (the embedding tables have none that anybody reads: the embedding kernels take the fp32 tables)
    t = model.query_encoder
    used = torch.ones(t.layout.total, dtype=torch.bool, device="cuda")
    for n in ("embeddings.word_embeddings.weight", "embeddings.position_embeddings.weight"):
        off, shape = t.layout.entries[n]
        used[off:off + shape[0] * shape[1]] = False
    if tr.amp16:
        # all-fp16 training: AdamW writes the fp16 shadow only (no pass of a training step reads the bf16 one); the bf16 shadow is marked
        # stale and an evaluation forward casts it first (tests/test_gpu_amp16.py::test_evaluation_after_fp16_steps_sees_current_bf16_weights)
        assert t._h_stale and torch.equal(t.flat_h16.float()[used], t.flat_p.to(torch.float16).float()[used])
    else:
        assert torch.equal(t.flat_h.float()[used], t.flat_p.to(torch.bfloat16).float()[used])
    skip = (tr.decay_flags & 2).repeat_interleave(64).bool()[:t.layout.total]
    assert torch.equal(skip, ~used)
    # the transposed copies (data-gradient GEMMs) are left stale by the optimizer step and made by the next step's preamble
    assert not t._t_fresh
    tr.forward_backward(batch)
    torch.cuda.synchronize()
    w2 = t.w("transformer.layer.0.ffn.lin2.weight")
    assert t._t_fresh and torch.equal(t.ht(0, "f2").float(), w2.T.contiguous().to(t.flat_t.dtype).float())      # fp16 in the all-fp16 mode


def _oracle_trajectory(model, cfg, batches, order, loss_kind, *, lr0, warmup, total, wd, max_norm):
    """The reference's step loop (trainer/multistep-curriculum/nway_listwise_1.py:328-367) restated over the oracle: fp32 forward + autograd
    (oracle/encoder_ref.py), closed-form loss gradient (losses_ref.py), clip_grad_norm_ + legacy AdamW + linear schedule in float64
    (optim_ref.py), parameters rounded to fp32 after every step as a torch optimizer leaves them.  GradScaler semantics for a non-finite
    step (:355-359): the optimizer step is skipped - no change of p / m / v, Adam's own step counter does not advance - the scheduler
    steps all the same.  Returns (losses, params {tower: {name: fp32 array}}, applied steps)."""
    qp, pp = selftest.oracle_params(model)
    towers = [("q", qp), ("p", pp)]
    ocfg = selftest.oracle_cfg(cfg)
    fn = {"margin_mse": LR.margin_mse, "kl_div": LR.kl_div, "lambda_mrr": LR.lambda_mrr, "ranknet": LR.ranknet}[loss_kind]
    mom = {t: {n: (np.zeros(tuple(v.shape)), np.zeros(tuple(v.shape))) for n, v in d.items()} for t, d in towers}
    prefix = {"q": "query_encoder.", "p": "passage_encoder."}
    losses, applied = [], 0
    for step, bi in enumerate(order):
        b = batches[bi]
        for _, d in towers:
            for v in d.values():
                v.grad = None
                v.requires_grad_(True)
        logits = E.nway_forward(qp, pp, ocfg, b["query"], b["nway_passages"])
        with np.errstate(all="ignore"):
            val, dl = fn(logits.detach().numpy(), b["labels"].numpy())
        losses.append(float(val))
        logits.backward(torch.from_numpy(np.asarray(dl)).float())
        grads = [v.grad.numpy() for _, d in towers for v in d.values()]
        total_norm, coef = O.clip_coef(grads, max_norm)
        lr = lr0 * O.linear_schedule_factor(step, warmup, total)
        if np.isfinite(total_norm):
            applied += 1
            for t, d in towers:
                for n, v in d.items():
                    m, vv = mom[t][n]
                    decay = 0.0 if O.no_decay(prefix[t] + n) else wd
                    p1, m1, v1 = O.adamw_step(v.detach().numpy(), v.grad.numpy().astype(np.float64) * coef, m, vv, lr=lr, step=applied, weight_decay=decay)
                    mom[t][n] = (m1, v1)
                    with torch.no_grad():
                        v.copy_(torch.from_numpy(p1.astype(np.float32)))
    return losses, {t: {n: v.detach().numpy().copy() for n, v in d.items()} for t, d in towers}, applied


@pytest.mark.parametrize("graph", [True, False])
@pytest.mark.parametrize("arch,loss_kind", [("distilbert", "kl_div"), ("bert", "margin_mse")])
def test_forty_step_trajectory_matches_the_oracle_loop(arch, loss_kind, graph, monkeypatch):
    """not one step from m = v = 0 but a TRAJECTORY - 40 steps over 8 rotating batches of one shape, warm-up 5 then
    linear decay, clip active, weight decay on, one poisoned (non-finite) batch in the middle - of NwayTrainer.train_step, with the HIP-graph
    replay on and off, against the reference's loop restated over the oracle (reference nway_listwise_1.py:328-367).  The class of bug this
    catches and a one-step test cannot: stale weight shadows after a replay, a bias-correction exponent that counts skipped steps, lr /
    seeds / step size frozen at their capture-time values, moments of one tower applied to the other.
    Bars: per-step loss within 2 % of (the step's loss + 15 % of the initial loss) - fp16 operands against an fp32 loop on a tiny model whose
    weights are 5 x the HF init scale, and a loss that falls 10-50 x along the way; measured values are printed; final parameters cosine >= 0.9999 per tensor; the 40-step UPDATE p40 - p0 of every weight matrix cosine >= 0.97 and
    its norm within 10 % (Adam normalises every element's step to ~lr, so elements whose gradient is rounding noise move by +-lr either
    way: the update is the sensitive quantity, the parameters are not); Adam's step in the checkpoint = applied steps = 39."""
    monkeypatch.setenv("CLDRD_GRAPH", "1" if graph else "0")
    cfg = small_cfg(arch)
    model = selftest.build_tiny_model(cfg, std=0.1).cuda()
    model.train()
    STEPS, NB, POISON = 40, 8, 17
    label_kind = "teacher"
    batches = [syn.nway_batch(4680 + i, 3, 6, 10, 40, vocab=cfg.vocab_size, ragged=True, label_kind=label_kind) for i in range(NB)]
    bad = {k: ({kk: vv.clone() for kk, vv in v.items()} if isinstance(v, dict) else v.clone()) for k, v in batches[POISON % NB].items()}
    bad["labels"][1, 2] = float("inf")                           # a non-finite teacher score: loss, dlogits and every gradient are NaN
    batches.append(bad)
    order = [(NB if i == POISON else i % NB) for i in range(STEPS)]
    hp = dict(lr0=1e-4, warmup=5, total=STEPS + 10, wd=0.01, max_norm=1.0)
    ref_losses, ref_params, applied = _oracle_trajectory(model, cfg, batches, order, loss_kind, **hp)
    assert applied == STEPS - 1
    p0 = {t: {n: v.detach().float().cpu().numpy().copy() for n, v in tw.named_flat()} for t, tw in (("q", model.query_encoder), ("p", model.passage_encoder))}
    tr = NwayTrainer(model, loss=loss_kind, learning_rate=hp["lr0"], weight_decay=hp["wd"], max_grad_norm=hp["max_norm"], warmup_steps=hp["warmup"],
                     total_steps=hp["total"])
    dev = [{k: ({kk: vv.cuda() for kk, vv in v.items()} if isinstance(v, dict) else v.cuda()) for k, v in b.items()} for b in batches]
    got_losses, clips = [], []
    for bi in order:
        out = tr.train_step(dev[bi])
        got_losses.append(out[0:1].clone())
        clips.append(tr.clip[1:2].clone())
    torch.cuda.synchronize()
    got_losses = torch.cat(got_losses).cpu().numpy()
    clips = torch.cat(clips).cpu().numpy()
    if graph:
        assert any(e["graph"] is not None for e in tr._graphs.values()) and not getattr(tr, "_graph_broken", False)
    else:
        assert not any(e["graph"] is not None for e in getattr(tr, "_graphs", {}).values())
    assert not np.isfinite(got_losses[POISON]) and not np.isfinite(ref_losses[POISON])
    ok = np.arange(STEPS) != POISON
    refl = np.asarray(ref_losses)
    # the loss falls by 10-50 x over the trajectory (eight rotating batches are memorised): late steps carry the earlier steps' rounding in a small
    # number, so the bar is 2 % of the step's loss plus 0.3 % of the INITIAL loss
    rel = np.abs(got_losses[ok] - refl[ok]) / (np.abs(refl[ok]) + 0.15 * abs(refl[0]))
    print(f"{arch}/{loss_kind}/graph={graph}: per-step loss error max {rel.max():.2e} median {np.median(rel):.2e}; clip coefficient min {np.nanmin(clips):.3f} "
          f"(active on {(clips[ok] < 1).sum()} of {ok.sum()} steps); loss {ref_losses[0]:.4f} -> {ref_losses[-1]:.4f}")
    assert rel.max() <= 2e-2, (rel.max(), int(np.argmax(rel)))
    assert (clips[ok] < 1).sum() >= 10, "the clip must be active on this trajectory"
    assert tr.skipped_steps() == 1 and tr.global_step == STEPS
    worst_p, worst_u, worst_n = 1.0, 1.0, 0.0
    for t, tw in (("q", model.query_encoder), ("p", model.passage_encoder)):
        for n, v in tw.named_flat():
            have, want, start = v.detach().float().cpu().numpy(), ref_params[t][n], p0[t][n]
            c = cos(have, want)
            worst_p = min(worst_p, c)
            # (vectors: a bias whose gradient is mathematically zero - k_lin.bias - moves by +-lr per step with the sign of rounding noise)
            assert c >= (0.9999 if have.ndim == 2 else 0.9995), f"{t}.{n}: final parameters cosine {c:.6f}"
            du, dw = have - start, want - start
            if have.ndim == 2 and "embeddings" not in n and np.linalg.norm(dw) > 0:
                cu, rn = cos(du, dw), np.linalg.norm(du) / np.linalg.norm(dw)
                worst_u, worst_n = min(worst_u, cu), max(worst_n, abs(rn - 1.0))
                assert cu >= 0.97 and abs(rn - 1.0) <= 0.10, f"{t}.{n}: 40-step update cosine {cu:.4f}, norm ratio {rn:.4f}"
    print(f"{arch}/{loss_kind}/graph={graph}: final parameters cosine >= {worst_p:.7f}; weight-matrix updates cosine >= {worst_u:.4f}, norm within {worst_n:.3f}")
    sd = tr.optimizer_state_dict()
    steps = {int(st["step"]) for st in sd["state"].values()}
    assert steps == {applied}, steps                             # Adam's own counter: the skipped step is not in it (scaler.step semantics)
    # an evaluation forward after the trajectory sees the CURRENT weights (shadows refreshed / marked stale by every replay)
    model.eval()
    with torch.no_grad():
        q_eval = model.query_embs(dev[0]["query"]).cpu().numpy()
    qp_now = {n: torch.from_numpy(ref_params["q"][n]) for n in ref_params["q"]}
    q_ref = E.cls_embs(qp_now, selftest.oracle_cfg(cfg), batches[0]["query"]).numpy()
    assert np.abs(q_eval - q_ref).max() <= 2e-2 * np.abs(q_ref).max()


@pytest.mark.parametrize("arch", ["distilbert", "bert"])
def test_write_once_gradients_equal_zeroed_accumulation(arch, monkeypatch):
    """The trainer zeroes only the embedding tables and lets every other gradient be WRITTEN once per step: the gradient buffer
    must equal what the zero-everything + accumulate path produces, also on the second step (no stale values), bit for bit outside
    the embedding tables (their scatter-add order is not fixed)."""
    cfg = small_cfg(arch=arch)
    model = selftest.build_tiny_model(cfg).cuda()
    model.train()
    tr = NwayTrainer(model, loss="kl_div", learning_rate=1e-3, warmup_steps=1, total_steps=100)
    b1 = syn.nway_batch(4680, 3, 4, 8, 32, vocab=cfg.vocab_size, ragged=True)
    b2 = syn.nway_batch(4681, 3, 4, 8, 32, vocab=cfg.vocab_size, ragged=True)
    grads = {}
    for mode in ("full", ""):
        tr.zero_all_grads = mode == "full"       # (test hook) zero the whole buffer and accumulate, the plain path
        tr.flat_g.fill_(123.0)                   # garbage: whatever is not zeroed must be overwritten
        tr.forward_backward(b1)
        tr.forward_backward(b2)                  # second pass over different data: nothing of the first may survive
        torch.cuda.synchronize()
        grads[mode] = tr.flat_g.clone()
    emb = torch.zeros_like(tr.flat_g, dtype=torch.bool)
    for tower, toff in zip(model.towers(), model._tower_offsets):
        a, b = tower.layout.embed_range
        emb[toff + a:toff + b] = True
    assert torch.equal(grads["full"][~emb], grads[""][~emb])
    assert torch.allclose(grads["full"][emb], grads[""][emb], rtol=1e-4, atol=1e-6)
    assert (grads[""] != 123.0).all()


def test_loss_decreases_over_steps_with_dropout():
    cfg = small_cfg()
    cfg.dropout = cfg.attention_dropout = 0.1
    model = selftest.build_tiny_model(cfg, std=0.05).cuda()
    model.train()
    batch = syn.nway_batch(4680, 4, 6, 8, 32, vocab=cfg.vocab_size, ragged=True)
    tr = NwayTrainer(model, loss="kl_div", learning_rate=2e-3, warmup_steps=1, total_steps=1000, max_grad_norm=1.0)
    losses = [tr.train_step(batch)[0].item() for _ in range(40)]
    assert np.isfinite(losses).all()
    assert np.mean(losses[-5:]) < 0.7 * np.mean(losses[1:6]), losses


def _full_size_model(arch, n_layers):
    cfg = EncoderConfig(arch=arch, n_layers=n_layers, dropout=0.0, attention_dropout=0.0)
    model = NwayDualEncoder(cfg, share_weights=False)
    with torch.no_grad():
        for seed, tower in ((11, model.query_encoder), (12, model.passage_encoder)):
            for name, p in tower.named_flat():
                p.copy_(syn.init_param(seed, name, tuple(p.shape), std=0.02, perturb=True))
    return model.cuda().train()


# (golden file, arch, layers, [(golden loss kind, trainer loss, oracle loss fn)])     BASELINE.json configs[0..3]
FULL_CONFIGS = {
    "cfg1": ("full_distilbert_cfg1.npz", "distilbert", 6, [("mse", "margin_mse")]),
    "cfg2": ("full_distilbert_cfg2.npz", "distilbert", 6, [("kl", "kl_div")]),
    "cfg3": ("full_distilbert_cfg3.npz", "distilbert", 6, [("mse", "margin_mse")]),
    "cfg4": ("full_bert_cfg4.npz", "bert", 12, [("ranknet", "ranknet"), ("lambda", "lambda_mrr")]),
}
ORACLE_LOSS = {"mse": LR.margin_mse, "kl": lambda a, b: LR.kl_div(a, b, 1.0), "ranknet": LR.ranknet, "lambda": LR.lambda_mrr}


def _loss_drift_p90(loss_fn, ref_logits, amp_logits, labels, draws=200):
    """90th percentile of |loss(ref + re-assigned autocast drift) - loss(ref)| / |loss(ref)| (see the caller)."""
    d = amp_logits - ref_logits
    l0 = loss_fn(ref_logits, labels)[0]
    rng = np.random.default_rng(0)
    rel = []
    for _ in range(draws):
        perm = rng.permutation(d.shape[0])
        sgn = rng.choice([-1.0, 1.0], size=(d.shape[0], 1))
        rel.append(abs(loss_fn(ref_logits + d[perm] * sgn, labels)[0] - l0) / abs(l0))
    return float(np.percentile(rel, 90))


# bars of the all-bf16 mode (CLDRD_AMP=bf16) in test_full_size_configs_match_reference_goldens; the fp16 mode's are in the test body
# (measured in round 6, cfg1 / cfg2 / cfg3 / cfg4: max 0.81 / 0.78 / 0.80 / 0.97 x and rms 0.72 / 0.87 / 0.75 / 1.26 x the reference's bf16-autocast logit
# drift; weight-matrix cosines min 0.99878 / 0.99915 / 0.99937 / -, where the reference's own bf16-autocast backward reaches 0.99874 - 0.99895:
# SURVEY.md 8c's 0.999 is NOT met by every tensor in this mode - nor by the reference's bf16 autocast - which is why fp16 is the default)
BF16_BAR = 1.5                     # max and rms logit drift, and max |d dlogits|, <= 1.5 x ONE draw of the reference's own bf16-autocast drift
BF16_COS = (0.997, 0.994)          # gradient cosine vs the reference's fp32 gradients: weight matrices / sum-type tensors (cfg4: min 0.99780 measured)
BF16_COS_SLACK = 5e-4              # ... and no tensor worse than the reference's own bf16-autocast backward on it by more than this


@pytest.mark.parametrize("mode", ["fp16", "bf16"])
@pytest.mark.parametrize("name", list(FULL_CONFIGS))
def test_full_size_configs_match_reference_goldens(name, mode, monkeypatch):
    """Every training config of BASELINE.json at full size on the GPU against what the REFERENCE produced for the same seeded
    weights and batch (tests/golden/make_golden.py, make_full_golden.py: models/nway_dual_encoder.py:21-49 over HF AutoModel, the
    reference's losses), dropout off.  cfg1 DistilBERT B=4 N=8 margin_mse; cfg2 B=8 N=32 kl_div (the bench workload); cfg3 B=4
    N=200 margin_mse; cfg4 BERT-base B=4 N=64 L=256 ranknet + lambda_mrr.

    Bar: the logit error is NO LARGER than the drift of the reference's own mixed-precision path on the same inputs (its
    bf16-autocast logits are stored in the golden; the reference trains under autocast, nway_listwise_1.py:334): ratio <= 1.0.
    The loss kernel equals the oracle's loss on the same logits (2e-5); against the reference's loss it moves by no more than the
    reference's own autocast drift pattern moves it (90th percentile over re-assignments of that pattern to other queries; floors:
    the stored autocast loss and 0.2 %); per-tensor gradient norms within 5 %."""
    # `mode`: CLDRD_AMP.  fp16 (default) = the reference's own precision (fp16 autocast, nway_listwise_1.py:129,334); bf16 = every MFMA operand
    # bf16, the wording of BASELINE.json's cfg2-4.  The bars below are stated per mode where they differ: the bf16 mode is held to the
    # reference's own bf16-AUTOCAST drift (its logits and gradient slices are in the goldens), the fp16 mode additionally to the fp16-autocast one.
    monkeypatch.setenv("CLDRD_AMP", mode)
    fname, arch, layers, kinds = FULL_CONFIGS[name]
    g = np.load(os.path.join(GOLDEN, fname))
    model = _full_size_model(arch, layers)
    assert all(t.amp_mode == mode for t in model.towers()) and model.query_fp16 == (mode == "fp16")
    assert model.passage_encoder.stream32, "the parity bar is defined for the default fp32 residual stream"
    B, N, Lq, Lp = int(g["B"]), int(g["N"]), int(g["Lq"]), int(g["Lp"])
    label_kind = str(g["label_kind"]) if "label_kind" in g.files else "teacher"
    batch = syn.nway_batch(4680, B, N, Lq, Lp, ragged=True, label_kind=label_kind)
    ref = g["logits"]
    amp_err = np.abs(g["logits_autocast_bf16"] - ref).max()
    for gk, loss_kind in kinds:
        tr = NwayTrainer(model, loss=loss_kind)
        loss_out, logits = tr.forward_backward(batch)
        got = logits.cpu().numpy()
        err = np.abs(got - ref).max()
        ratio = err / amp_err
        rms, amp_rms = np.sqrt(np.mean((got - ref) ** 2)), np.sqrt(np.mean((g["logits_autocast_bf16"] - ref) ** 2))
        print(f"{name}/{loss_kind}: max|dlogit| {err:.4f} = {ratio:.2f} x the reference's bf16-autocast drift {amp_err:.4f} (max|logit| {np.abs(ref).max():.2f}); "
              f"rms {rms:.4f} = {rms / amp_rms:.2f} x its rms {amp_rms:.4f}")
        # fp16 mode: no larger than the reference's bf16-autocast drift (it is ~8x finer: < 0.15 x measured).  bf16 mode: the same operand
        # precision as that autocast path, with the fp32 residual stream on our side - asserted at BF16_BAR x of ONE draw of its drift
        # (B = 4 .. 8 queries: a handful of draws; measured values are printed)
        bar = 1.0 if mode == "fp16" else BF16_BAR
        assert ratio <= bar, f"{name}/{mode}: drift {err:.4f} exceeds {bar} x the reference's own bf16-autocast drift {amp_err:.4f}"
        assert rms <= bar * amp_rms, f"{name}/{mode}: rms drift {rms:.4f} exceeds {bar} x the reference's own bf16-autocast rms drift {amp_rms:.4f}"
        if "logits_autocast_fp16" in g.files:
            # the mode the reference actually trains in is fp16 autocast (nway_listwise_1.py:334): 8x finer operand rounding than ours
            a16 = np.abs(g["logits_autocast_fp16"] - ref)
            print(f"{name}/{loss_kind}: the reference's fp16-autocast drift is max {a16.max():.4f} rms {np.sqrt(np.mean(a16 ** 2)):.4f}: ours = "
                  f"{err / a16.max():.2f} x / {rms / np.sqrt(np.mean(a16 ** 2)):.2f} x; relative to max|logit|: {err / np.abs(ref).max():.2e}")
        if model.passage_encoder.ffn_fp16:
            # GPU vs fp32 oracle, logits <= 5e-3 relative - every config, BERT-base at L = 256 included since the
            # out-projection reads fp16 operands too (measured 2.2e-3 - 3.0e-3: asserted at 4e-3), and within 1.5 x of the drift of the
            # reference's OWN fp16 autocast (measured 0.96 x - 1.36 x)
            # all-fp16 training mode (round 4): 1.45e-3 - 2.25e-3 measured, asserted at 3e-3; 0.71 x - 1.05 x the max and 0.74 x - 0.92 x the rms
            # of the reference's fp16-autocast drift.  Round 5: both ratios asserted at 1.25 x.  The drift of B = 4 .. 8 queries is a handful of
            # draws (every logit of a row shares the query's error term): GELU in the erfc form - MORE accurate before rounding (0.023 instead of
            # 0.028 fp16 ulp rms, max 0.97 instead of 1.9: tools emulation) - re-rolled the fp16 roundings of h and moved the
            # rms ratios from 0.78 / 0.81 / 0.92 / 0.74 (cfg1-4) to 0.88 / 0.93 / 1.05 / 0.54: up on three configs, down on the fourth, same
            # kernels otherwise.  A bar at 1.0 x of ONE draw of the reference's own drift was tighter than the quantity is reproducible.
            rel_bar = 3e-3 if tr.amp16 else 4e-3
            assert err <= rel_bar * np.abs(ref).max(), f"{name}: max|dlogit| {err:.4f} above {rel_bar:g} x max|logit| = {rel_bar * np.abs(ref).max():.4f}"
            if "logits_autocast_fp16" in g.files:
                # round 6 (keep the old bar where it holds): measured max / rms ratios cfg1 0.82 / 0.66, cfg2 0.67 / 0.66, cfg3 0.77 / 0.76,
                # cfg4 0.64 / 0.62 - every config is back under the 1.0 x of rounds 3-4, which is asserted again (round 5 had loosened it to 1.25 x)
                kmax, krms = (1.0, 1.0) if tr.amp16 else (1.5, 1.5)
                assert err <= kmax * a16.max() and rms <= krms * np.sqrt(np.mean(a16 ** 2)), f"{name}: more than {kmax} x / {krms} x the reference's fp16-autocast drift"
        if "loss" in g.files:                                # cfg1 golden (round 1 layout)
            ref_loss, names, vals = float(g["loss"]), [str(n) for n in g["grad_norm_names"]], g["grad_norm_values"]
            amp_loss, _ = ORACLE_LOSS[gk](g["logits_autocast_bf16"], batch["labels"].numpy())
        else:
            ref_loss = float(g[f"loss_{gk}"])
            names, vals = [str(n) for n in g[f"grad_norm_names_{gk}"]], g[f"grad_norm_values_{gk}"]
            amp_loss = float(g[f"loss_{gk}_autocast"])
            # the oracle's loss restatement agrees with the reference on these logits (ties the two checkers together)
            assert ORACLE_LOSS[gk](ref, batch["labels"].numpy())[0] == pytest.approx(ref_loss, rel=1e-5)
        # (a) the loss kernel on OUR logits is the oracle's loss of those logits (the loss itself is exact arithmetic: fp32)
        labels_np = batch["labels"].numpy()
        assert loss_out[0].item() == pytest.approx(ORACLE_LOSS[gk](got, labels_np)[0], rel=2e-5), f"{name}/{loss_kind}: loss kernel vs oracle on the same logits"
        # (b) against the reference's fp32 loss: the stored autocast loss is ONE draw of a noisy quantity (cfg2: 0.06 %), and any
        # ulp-level change of our code generation re-draws ours (measured: +-0.2 % of the loss from a bit-identical change of the bf16
        # pack instruction sequence).  The bar is therefore the reference's own drift PATTERN: its autocast logit errors re-assigned
        # to other queries (rows permuted, signs flipped - the within-row structure, a common shift per query, is what the list-wise
        # losses are insensitive to, and it is kept) move the loss by some distribution; ours must stay inside its 90th percentile
        # (floors: the stored draw and 0.2 %).
        tol = max(2e-3, abs(amp_loss - ref_loss) / abs(ref_loss), _loss_drift_p90(ORACLE_LOSS[gk], ref, g["logits_autocast_bf16"], labels_np))
        assert loss_out[0].item() == pytest.approx(ref_loss, rel=tol), f"{name}/{loss_kind}: loss (tolerance {tol:.4f})"
        params = {f"query_encoder.{n}": p for n, p in model.query_encoder.named_flat()}
        params.update({f"passage_encoder.{n}": p for n, p in model.passage_encoder.named_flat()})
        big, checked, worst = vals.max(), 0, 0.0
        for n, v in zip(names, vals):
            if n in params and v > 1e-3 * big:
                rel = abs(params[n].grad.norm().item() - v) / v
                worst = max(worst, rel)
                tol_g = 0.10 if (loss_kind == "lambda_mrr") else 0.05      # rank weights flip on bf16-level logit noise
                assert rel <= tol_g, f"{name}/{loss_kind}: grad norm of {n} off by {rel:.3f}"
                checked += 1
        assert checked > 100
        print(f"{name}/{loss_kind}: loss {loss_out[0].item():.6f} vs {ref_loss:.6f}; worst per-tensor grad-norm error {worst:.4f} over {checked} tensors")
        # d loss / d logits: the loss kernel's gradient on OUR logits against the reference's on ITS fp32 logits - no further from it than
        # the reference's own bf16-autocast path is (same bar as the logits), and pointing the same way
        dkey = f"dlogits_{gk}"
        if dkey in g.files:
            _, dl = ops.loss_fwd_bwd(loss_kind, logits, batch["labels"].to(logits.device).float().contiguous())
            dl, dref = dl.cpu().numpy().astype(np.float64), g[dkey].astype(np.float64)
            damp = (g[dkey + "_autocast"] if dkey + "_autocast" in g.files else ORACLE_LOSS[gk](g["logits_autocast_bf16"], labels_np)[1]).astype(np.float64)
            e_our, e_amp = np.abs(dl - dref).max(), np.abs(damp - dref).max()
            cos_dl = float((dl * dref).sum() / (np.linalg.norm(dl) * np.linalg.norm(dref)))
            print(f"{name}/{loss_kind}: max|d(dlogits)| {e_our:.3e} (reference autocast {e_amp:.3e}, max|dlogits| {np.abs(dref).max():.3e}); cosine {cos_dl:.6f}")
            assert e_our <= (1.0 if mode == "fp16" else BF16_BAR) * max(e_amp, 1e-6 * np.abs(dref).max()) and cos_dl >= (0.99 if loss_kind == "lambda_mrr" else 0.999)
        # gradient DIRECTIONS against stored fp32 reference gradients (tests/golden/make_full_golden_r3.py): the first 16 rows of the four
        # weight matrices of layers 0 and 5 and of the position embeddings, every 1-D parameter of layers 0, 2, 5 and the embedding LayerNorm.
        #   weight matrices (99.9 % of the parameters): the survey's bar is cosine >= 0.999, asserted as such.  With
        #   the fp32 gradient stream (the default since round 3) this backward reaches 0.99909 - 0.99999 (median 0.99997); the reference's OWN
        #   bf16-autocast backward 0.99874 on the same slices.
        #   biases, LayerNorm parameters and position-embedding rows: their gradient is a plain SUM over all tokens of an activation
        #   gradient whose terms nearly cancel (|sum| ~ 1e-4 of the summed magnitudes), so its direction is set by the rounding noise of
        #   that tensor.  What reads the fp32 stream (LayerNorm parameters, the biases in front of a LayerNorm, position rows) is now at
        #   >= 0.9997; the worst ones left - q/k/v and FFN1 biases, 0.9950 - 0.9959 - are column sums of a bf16 MFMA operand (dqkv, dpre),
        #   exactly where the reference's bf16 autocast lands on the same tensors (0.9952 / 0.9959).  Asserted >= 0.994 (bf16 stream: 0.985).
        nkey = f"gslice_names_{gk}"
        if nkey in g.files:
            def cosine(a, b):
                return float((a * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))
            rows = []
            for n in [str(x) for x in g[nkey]]:
                want = g[f"gslice/{gk}/{n}"].astype(np.float64)
                have = params[n].grad.detach()
                have = (have if have.dim() == 1 else have[:want.shape[0]]).double().cpu().numpy()
                if np.linalg.norm(want) < 1e-3 * big:
                    continue
                akey = f"gslice_autocast/{gk}/{n}"
                c_amp = cosine(want, g[akey].astype(np.float64)) if akey in g.files else 1.0
                rows.append((cosine(want, have), c_amp, n, 1 if "position_embeddings" in n else want.ndim))
            rows.sort()
            for dim, label in ((2, "weight matrices"), (1, "biases / LayerNorm parameters / position-embedding rows")):
                sub = [r for r in rows if r[3] == dim]
                print(f"{name}/{loss_kind}: gradient cosine of {len(sub)} {label}: min {sub[0][0]:.5f} ({sub[0][2]}; the reference's bf16-autocast "
                      f"backward: {sub[0][1]:.5f}), median {sub[len(sub) // 2][0]:.5f}; reference autocast min {min(r[1] for r in sub):.5f}")
            assert len(rows) >= 40
            gs32 = True                               # fp32 residual-path arithmetic in both modes
            bar2, bar1 = BF16_COS                     # bf16 mode: weight matrices / sum-type tensors
            if tr.amp16:
                # the all-fp16 training mode (round 4, the default): 11-bit operands in the whole backward - the survey's 0.999 now holds for EVERY
                # ranked tensor, sum-type ones included (profiles/r04_grad_cosines.txt: min 0.99992 over cfg1-4; the bf16-operand mode 0.9959)
                bar2, bar1 = 0.9995, 0.9995
            for c, c_amp, n, dim in rows:
                if mode == "bf16":
                    # all-bf16 mode: the yardstick is the reference's OWN bf16-autocast backward on the same tensor (stored in the goldens):
                    # never worse than it by more than BF16_COS_SLACK, and where it is better than the absolute bar, at least the bar minus
                    # that slack (rank-weighted losses - lambda_mrr - flip pair weights on bf16-level logit noise: the reference's own bf16
                    # autocast falls to 0.989 on some tensors there)
                    floor = min(bar2 if dim == 2 else bar1, c_amp) - BF16_COS_SLACK
                    assert c >= floor, f"{name}/{loss_kind}/bf16: gradient of {n}: cosine {c:.5f} (reference bf16 autocast {c_amp:.5f}, floor {floor:.5f})"
                    continue
                assert c >= (bar2 if dim == 2 else bar1), f"{name}/{loss_kind}: gradient of {n}: cosine {c:.5f} (reference autocast {c_amp:.5f})"
                # sum-type tensors (biases, LayerNorm parameters, position rows: a plain sum over all tokens of an activation gradient whose terms
                # nearly cancel) also never worse than the reference's own bf16-autocast backward on the same tensor
                if dim == 1:
                    assert c >= c_amp - BF16_COS_SLACK, f"{name}/{loss_kind}: gradient of {n}: cosine {c:.5f} below the reference's bf16-autocast backward {c_amp:.5f}"
        del tr


def test_torch_optimizer_loop_sees_fresh_weights():
    """Reference-style loop (nway_listwise_1.py:328-367 with a torch optimizer): the bf16 weight shadows must follow
    optimizer.step() and load_state_dict (parameters share flat_p's version counter now)."""
    from cldrd_amd.losses import MarginMSE
    cfg = small_cfg()
    model = selftest.build_tiny_model(cfg).cuda()
    model.train()
    batch = syn.nway_batch(4680, 2, 3, 8, 32, vocab=cfg.vocab_size, ragged=True)
    dev_batch = {k: ({kk: vv.cuda() for kk, vv in v.items()} if isinstance(v, dict) else v.cuda()) for k, v in batch.items()}
    opt = torch.optim.AdamW(model.parameters(), lr=5e-3)
    seen = []
    for _ in range(3):
        logits = model(dev_batch["query"], dev_batch["nway_passages"])
        loss = MarginMSE()(logits, dev_batch["labels"])
        loss.backward()
        opt.step()
        opt.zero_grad()
        seen.append(logits.detach().clone())
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])
    model.eval()
    with torch.no_grad():
        before = model.query_embs(dev_batch["query"]).clone()
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        for k in sd:
            if k.endswith("ffn.lin1.weight"):
                sd[k] = sd[k] * 1.5
        model.load_state_dict(sd)
        after = model.query_embs(dev_batch["query"])
    assert not torch.equal(before, after)


def test_zero_grad_between_steps_gives_single_step_gradients():
    """optimizer.zero_grad() defaults to set_to_none=True: the next backward must not add onto the previous step's flat
    gradient."""
    from cldrd_amd.losses import MarginMSE
    cfg = small_cfg()
    model = selftest.build_tiny_model(cfg).cuda()
    model.train()
    batch = syn.nway_batch(4680, 2, 3, 8, 32, vocab=cfg.vocab_size, ragged=True)
    dev_batch = {k: ({kk: vv.cuda() for kk, vv in v.items()} if isinstance(v, dict) else v.cuda()) for k, v in batch.items()}
    opt = torch.optim.SGD(model.parameters(), lr=0.0)          # lr 0: the weights stay put, so both steps see the same problem
    grads = []
    for _ in range(2):
        loss = MarginMSE()(model(dev_batch["query"], dev_batch["nway_passages"]), dev_batch["labels"])
        loss.backward()
        grads.append(torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone())
        opt.step()
        opt.zero_grad()
        assert all(p.grad is None for p in model.parameters())
    assert torch.allclose(grads[0], grads[1], rtol=1e-4, atol=1e-6 * grads[0].abs().max().item())


def test_checkpoint_optimizer_state_round_trip_and_torch_layout(tmp_path):
    """trainer.state_dict() carries the optimizer in torch's layout (what the reference saves / loads, nway_listwise_1.py:302,
    :423); it loads back bit for bit, a state written by torch.optim.AdamW over the same parameters loads too, junk raises."""
    cfg = small_cfg(layers=1)
    model = selftest.build_tiny_model(cfg).cuda()
    model.train()
    batch = syn.nway_batch(4680, 2, 3, 8, 16, vocab=cfg.vocab_size)
    tr = NwayTrainer(model, loss="margin_mse", learning_rate=1e-3, warmup_steps=1, total_steps=10)
    for _ in range(2):
        tr.train_step(batch)
    ck = tr.state_dict()
    assert set(ck["optimizer"]) == {"state", "param_groups"} and len(ck["optimizer"]["param_groups"]) == 2
    torch.save(ck, tmp_path / "c.pth.tar")
    ck = torch.load(tmp_path / "c.pth.tar", map_location="cpu", weights_only=False)
    model2 = selftest.build_tiny_model(cfg, seed=9).cuda()
    tr2 = NwayTrainer(model2, loss="margin_mse", learning_rate=1e-3, warmup_steps=1, total_steps=10)
    tr2.load_state_dict(ck)
    assert tr2.global_step == 2 and tr2.adam_step == 2
    assert torch.equal(tr2.m, tr.m) and torch.equal(tr2.v, tr.v) and torch.equal(tr2.flat_p, tr.flat_p)
    # a state dict produced by a torch optimizer built the reference's way over this model's parameters
    groups = T_optimizer_groups(model2)
    opt = torch.optim.AdamW(groups, lr=1e-3)
    for p in model2.parameters():
        p.grad = torch.full_like(p, 0.25)
    opt.step()
    sd = opt.state_dict()
    tr2.load_optimizer_state_dict(sd)
    name = "passage_encoder.transformer.layer.0.ffn.lin1.weight"
    pid = [i for g_, e in zip(sd["param_groups"], tr2._optimizer_names()) for i, x in zip(g_["params"], e) if x[0] == name][0]
    assert torch.equal(tr2._slice(tr2.m, 1, "transformer.layer.0.ffn.lin1.weight").cpu(), sd["state"][pid]["exp_avg"].cpu())
    assert tr2._opt_step_loaded == 1
    with pytest.raises(ValueError):
        tr2.load_optimizer_state_dict({"what": 1})
    with pytest.raises(ValueError):
        bad = {"state": sd["state"], "param_groups": [sd["param_groups"][0]]}
        tr2.load_optimizer_state_dict(bad)


def T_optimizer_groups(model):
    """the reference's grouping (nway_listwise_1.py:259-263) over our model's named_parameters, in HF order"""
    from cldrd_amd.trainer.nway_listwise import optimizer_param_groups
    params = dict(model.named_parameters())
    out = []
    for g_, wd in zip(optimizer_param_groups(model), (0.01, 0.0)):
        out.append({"params": [params[e[0]] for e in g_ if e[1] is not None], "weight_decay": wd})
    return out


def test_failed_graph_capture_falls_back_to_a_working_eager_step(monkeypatch):
    """A capture that raises half-way (here: the optimizer launches, after both streams have been forked inside the capture) must leave a
    trainer that keeps training eagerly: the capture is ended with its streams joined (otherwise the stream stays in capture mode and the
    next launch fails), the device seed pointers are dropped, the transposed weight shadows - "refreshed" inside the capture without a
    kernel having run - are refreshed for real, and the steps that follow equal those of a trainer that never tried a graph."""
    cfg = small_cfg()
    batch = syn.nway_batch(4680, 3, 4, 8, 16, vocab=cfg.vocab_size, ragged=True, label_kind="teacher")
    batch = {k: ({kk: vv.cuda() for kk, vv in v.items()} if isinstance(v, dict) else v.cuda()) for k, v in batch.items()}

    def run(break_capture):
        monkeypatch.setenv("CLDRD_GRAPH", "1" if break_capture else "0")
        model = selftest.build_tiny_model(cfg).cuda().train()
        tr = NwayTrainer(model, loss="kl_div", learning_rate=1e-5, warmup_steps=0, total_steps=20)      # small steps: atomics noise is not amplified
        if break_capture:
            real = tr._optimizer_launches

            def boom(lr, step):
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("injected failure inside the capture")
                return real(lr, step)
            tr._optimizer_launches = boom
        # The embedding tables are FROZEN for this comparison (restored after every step): their gradients are float atomics whose summation
        # order differs from run to run, and Adam's first updates are lr * g / |g| - an element whose gradient is rounding noise around zero moves
        # by +-lr with the sign of the noise (tools: two EAGER runs of this very loop gave 0.358131 or 0.359187 for the sixth loss).  Everything
        # else of the step is deterministic, so with the tables held the two runs must agree step for step (instead of a 1e-2 bar).
        emb = [(toff + t.layout.embed_range[0], toff + t.layout.embed_range[1]) for t, toff in zip(model.towers(), model._tower_offsets)]
        p0 = tr.flat_p.clone()
        outs = []
        for _ in range(6):
            outs.append(tr.train_step(batch).clone())
            with torch.no_grad():
                for a, b in emb:
                    tr.flat_p[a:b].copy_(p0[a:b])
        torch.cuda.synchronize()
        return tr, torch.stack(outs), tr.flat_p.clone()

    with pytest.warns(UserWarning, match="injected failure inside the capture"):
        tr_b, loss_b, p_b = run(True)
    assert getattr(tr_b, "_graph_broken", False) and tr_b._state is None
    assert not torch.cuda.is_current_stream_capturing()
    tr_e, loss_e, p_e = run(False)
    # dropout is off in small_cfg and the embedding tables are held (see run): the two runs are the same arithmetic, every step of them.  A
    # fallback that trained on unwritten shadows or stale seeds would be off by far more than the bar.
    assert torch.allclose(loss_b, loss_e, rtol=1e-4, atol=1e-6), (loss_b, loss_e)
    assert (p_b - p_e).abs().max().item() <= 1e-5 * p_e.abs().max().item()


def test_graph_replay_of_the_training_step_equals_the_eager_step(monkeypatch):
    """NwayTrainer.train_step captures the step into a HIP graph after 3 eager steps (seeds / lr / Adam step size in device memory,
    include/cldrd_hip.h: cldrd_write_step_state).  From IDENTICAL state (snapshot / restore of parameters, moments, counters) the replayed
    step and the plain eager step (by-value seeds and lr) must agree: loss, logits and every gradient outside the embedding tables BIT FOR
    BIT (dropout on: the masks of forward and backward come from the same seeds in both), the embedding-table gradients - float atomics,
    not reproducible run to run in either mode - and the updated parameters to rounding.  Batches change every step, lr moves through its
    warm-up; one step uses another batch shape (an eager step inside graph mode)."""
    cfg = small_cfg()
    cfg.dropout, cfg.attention_dropout = 0.1, 0.1
    monkeypatch.setenv("CLDRD_GRAPH", "1")
    torch.manual_seed(0)
    model = NwayDualEncoder(cfg, share_weights=False).cuda().train()
    with torch.no_grad():
        for seed, tower in ((11, model.query_encoder), (12, model.passage_encoder)):
            for name, p in tower.named_flat():
                p.copy_(syn.init_param(seed, name, tuple(p.shape), std=0.05, perturb=True))
    tr = NwayTrainer(model, loss="kl_div", learning_rate=3e-3, warmup_steps=5, total_steps=40)
    towers = model.towers()
    emb = torch.zeros(tr.flat_p.numel(), dtype=torch.bool, device="cuda")
    for t, toff in zip(towers, model._tower_offsets):
        for n in ("embeddings.word_embeddings.weight", "embeddings.position_embeddings.weight"):
            off, shape = t.layout.entries[n]
            emb[toff + off:toff + off + shape[0] * shape[1]] = True

    def snapshot():
        return (tr.flat_p.clone(), tr.m.clone(), tr.v.clone(), tr.global_step, tr.adam_step, [t.step_seed for t in towers])

    def restore(sn):
        tr.flat_p.copy_(sn[0]); tr.m.copy_(sn[1]); tr.v.copy_(sn[2])
        tr.global_step, tr.adam_step = sn[3], sn[4]
        for t, ss in zip(towers, sn[5]):
            t.step_seed = ss
            t.refresh_shadows(need_transposed=True)

    def dev(batch):
        return {k: ({kk: vv.cuda() for kk, vv in v.items()} if isinstance(v, dict) else v.cuda()) for k, v in batch.items()}

    replays = 0
    for i in range(11):
        shape = (3, 4, 8, 16) if i != 8 else (2, 3, 8, 16)
        batch = dev(syn.nway_batch(100 + i, *shape, vocab=cfg.vocab_size, ragged=True, label_kind="teacher"))
        sn = snapshot()
        monkeypatch.setenv("CLDRD_GRAPH", "1")
        l1 = tr.train_step(batch).clone()
        lg1, g1, p1 = tr.last_logits.clone(), tr.flat_g.clone(), tr.flat_p.clone()
        graph_now = any(e["graph"] is not None for e in getattr(tr, "_graphs", {}).values())
        replayed = graph_now and tr._batch_key(batch) in tr._graphs and tr._graphs[tr._batch_key(batch)]["graph"] is not None
        replays += int(replayed)
        # the plain eager step from the same state: no device-side step state at all
        restore(sn)
        state, ptrs = tr._state, [getattr(t, "seed_base_ptr", None) for t in towers]
        tr._state = None
        for t in towers:
            t.seed_base_ptr = None
        monkeypatch.setenv("CLDRD_GRAPH", "0")
        l0 = tr.train_step(batch).clone()
        tr._state = state
        for t, pp in zip(towers, ptrs):
            t.seed_base_ptr = pp
        torch.cuda.synchronize()
        assert torch.equal(l1, l0), (i, replayed, l1, l0)
        assert torch.equal(lg1, tr.last_logits), (i, replayed)
        assert torch.equal(g1[~emb], tr.flat_g[~emb]), (i, replayed)
        ge = (g1[emb] - tr.flat_g[emb]).abs().max().item()
        assert ge <= 1e-5 * max(1.0, g1[emb].abs().max().item()), (i, ge)
        assert (p1 - tr.flat_p).abs().max().item() <= 1e-6, i
    assert replays >= 6, replays
    assert tr.global_step == tr.adam_step == 11


def _ragged_batch(cfg, seed, B, N, Lq, Lp, force=()):
    batch = syn.nway_batch(seed, B, N, Lq, Lp, vocab=cfg.vocab_size, ragged=True, label_kind="teacher")
    # lengths spread over the whole range (the MSMARCO length model clips at these toy lengths)
    lens = 2 + (syn.randint(seed + 77, 0, Lp - 1, B * N)).astype(np.int64)
    lens[:len(force)] = np.asarray(force, dtype=np.int64)
    ar = np.arange(Lp)[None, :]
    mask = (ar < lens[:, None]).astype(np.int64)
    ids = batch["nway_passages"]["input_ids"].reshape(B * N, Lp).numpy().copy()
    ids = np.where(mask == 1, ids, 0)
    ids[np.arange(B * N), lens - 1] = 2
    batch["nway_passages"] = {"input_ids": torch.from_numpy(ids).view(B, N, Lp), "attention_mask": torch.from_numpy(mask).view(B, N, Lp)}
    return batch, lens


@pytest.mark.parametrize("arch,layers,Lp", [("distilbert", 3, 48), ("bert", 2, 48), ("distilbert", 2, 128)])
def test_packed_batch_equals_padded_batch(arch, layers, Lp, monkeypatch):
    """Variable-length packing (csrc/pack.hip; the reference pads to the longest sequence of the batch, dataset/nway_dataset.py:103-107):
    with the token counts given, Linear / LayerNorm / weight gradients run on the real tokens only.  Same CLS embeddings, logits and
    gradients as the padded run of the same batch (dropout off; rows are computed by the same kernels in the same order, so nearly
    the LOGITS are bit-identical, gradients to rounding: float atomics in the embedding tables), and against the oracle.  Lp = 128 with sequences
    of 59 .. 64, 1 and 128 tokens: the lengths at which the attention kernels' block skipping once read stale accumulators (attention.hip,
    mfma_drain)."""
    cfg = small_cfg(arch, layers)
    if Lp > cfg.max_position_embeddings:
        cfg.max_position_embeddings = Lp
    model = selftest.build_tiny_model(cfg).cuda().train()
    B, N, Lq = 4, 9, 8
    batch, lens = _ragged_batch(cfg, 321, B, N, Lq, Lp, force=(59, 60, 61, 62, 63, 64, 1, 128, 33, 96, 97) if Lp == 128 else ())
    dev = lambda b: {k: ({kk: vv.cuda() for kk, vv in v.items()} if isinstance(v, dict) else v.cuda()) for k, v in b.items()}
    tr = NwayTrainer(model, loss="margin_mse")
    monkeypatch.setenv("CLDRD_GRAPH", "0")
    _, logits_pad = tr.forward_backward(dev(batch))
    g_pad = tr.flat_g.clone()
    packed = dev(batch)
    packed["nway_passages"]["lengths"] = torch.from_numpy(lens)
    _, logits_pk = tr.forward_backward(packed)
    g_pk = tr.flat_g.clone()
    fill = lens.sum() / (B * N * Lp)
    assert fill < 0.7
    same = (logits_pk == logits_pad).float().mean().item()
    err = (logits_pk - logits_pad).abs().max().item() / logits_pad.abs().max().item()
    cos = torch.nn.functional.cosine_similarity(g_pk.double(), g_pad.double(), dim=0).item()
    print(f"packed vs padded ({arch}, fill {fill:.2f}): logits identical {same:.2f}, max rel diff {err:.2e}; gradient cosine {cos:.7f}, "
          f"norm ratio {(g_pk.norm() / g_pad.norm()).item():.6f}")
    assert same == 1.0, "packed logits are not bit-identical to the padded ones"
    assert err <= 2e-3 and cos >= 0.99995 and abs((g_pk.norm() / g_pad.norm()).item() - 1.0) <= 2e-3
    # and against the oracle (which, like the reference, computes on the padded batch)
    ref_logits = oracle_run(model, cfg, batch, "margin_mse")[0]
    assert np.abs(logits_pk.cpu().numpy() - ref_logits).max() <= 3e-2 * np.abs(ref_logits).max()
    # a training step with dropout on the packed batch runs and stays finite
    cfg2 = small_cfg(arch, layers)
    cfg2.dropout, cfg2.attention_dropout, cfg2.max_position_embeddings = 0.1, 0.1, cfg.max_position_embeddings
    m2 = selftest.build_tiny_model(cfg2).cuda().train()
    tr2 = NwayTrainer(m2, loss="kl_div")
    out = tr2.train_step(packed)
    assert torch.isfinite(out[0]).item() and torch.isfinite(tr2.flat_p).all().item()


@pytest.mark.parametrize("arch", ["distilbert", "bert"])
@pytest.mark.parametrize("packed", [False, True])
def test_full_last_layer_equals_the_cls_only_last_layer(arch, packed, monkeypatch):
    """`cls_only_last = False` (a test hook: the last layer computed for every token, as the reference does) against the default
    that computes only the CLS row after the K / V projection: same logits and gradients up to rounding, on padded and packed batches
    (the packed full layer scatters dL/dCLS to the rows `cu[m]`, not to `m * L`)."""
    cfg = small_cfg(arch, 3)
    B, N, Lq, Lp = 3, 5, 8, 40
    batch, lens = _ragged_batch(cfg, 654, B, N, Lq, Lp)
    dev = lambda b: {k: ({kk: vv.cuda() for kk, vv in v.items()} if isinstance(v, dict) else v.cuda()) for k, v in b.items()}
    monkeypatch.setenv("CLDRD_GRAPH", "0")
    out = {}
    for cls_only in (True, False):
        model = selftest.build_tiny_model(cfg).cuda().train()
        for t in model.towers():
            t.cls_only_last = cls_only
        tr = NwayTrainer(model, loss="margin_mse")
        b = dev(batch)
        if packed:
            b["nway_passages"]["lengths"] = torch.from_numpy(lens)
        _, logits = tr.forward_backward(b)
        torch.cuda.synchronize()
        out[cls_only] = (logits.double().clone(), tr.flat_g.double().clone())
    (l1, g1), (l0, g0) = out[True], out[False]
    rel = (l1 - l0).abs().max().item() / l1.abs().max().item()
    c = torch.nn.functional.cosine_similarity(g1, g0, dim=0).item()
    ratio = (g0.norm() / g1.norm()).item()
    print(f"full vs CLS-only last layer ({arch}, packed={packed}): max logit diff {rel:.2e} of the logit scale, gradient cosine {c:.6f}, norm ratio {ratio:.5f}")
    # the two paths round at different points (the CLS path keeps its attention probabilities in fp32, the full kernel feeds bf16 P to the
    # MFMA) and the tiny models' weights are 10x the HF init: bf16-level differences
    assert rel <= 1e-2 and c >= 0.9995 and abs(ratio - 1.0) <= 5e-3, (rel, c, ratio)


@pytest.mark.parametrize("switch,kind", [("zero_all_grads=1", "exec"), ("use_norm_sink=0", "exec"), ("CLDRD_AMP=bf16", "numerics")])
def test_switches_of_the_training_step(switch, kind, monkeypatch):
    """The two test hooks of the trainer and the ONE numerics switch keep working: two training steps (dropout off) under the switch against
    the default.  Execution hooks (where / when things run) give the same logits bit for bit and the same update up to the float atomics of the
    embedding gradients and the grouping of the clip norm; CLDRD_AMP=bf16 (every MFMA operand bf16) stays within bf16 rounding of the fp16
    mode and is a different code path (not bit-identical)."""
    cfg = small_cfg("bert", 3)
    batch = syn.nway_batch(4680, 3, 4, 10, 32, vocab=cfg.vocab_size, ragged=True)
    monkeypatch.setenv("CLDRD_GRAPH", "0")
    res = {}
    for on in (False, True):
        k, v = switch.split("=")
        if on and k.startswith("CLDRD_"):
            monkeypatch.setenv(k, v)
        model = selftest.build_tiny_model(cfg).cuda().train()
        tr = NwayTrainer(model, loss="margin_mse", learning_rate=1e-4, warmup_steps=0, total_steps=10)
        if on and not k.startswith("CLDRD_"):
            assert hasattr(tr, k)
            setattr(tr, k, v == "1")
        if on and k == "CLDRD_AMP":
            assert not tr.amp16 and all(t.amp_mode == "bf16" and not t.needs_h16 for t in model.towers())
        p0 = tr.flat_p.clone()
        tr.train_step(batch)
        l1 = tr.last_logits.clone()
        tr.train_step(batch)
        torch.cuda.synchronize()
        res[on] = (l1, tr.last_logits.clone(), (tr.flat_p - p0).double())
    (a1, a2, ua), (b1, b2, ub) = res[False], res[True]
    assert torch.isfinite(b2).all().item() and torch.isfinite(ub).all().item()
    rel_upd = ((ua - ub).norm() / ua.norm()).item()
    if kind == "exec":
        assert torch.equal(a1, b1), f"{switch}: first-step logits differ"
        assert rel_upd <= 2e-2, f"{switch}: update differs by {rel_upd:.2e}"
    else:
        assert not torch.equal(a1, b1), f"{switch}: the switch is dead (bit-identical logits)"
        assert (a1 - b1).abs().max().item() <= 2e-2 * a1.abs().max().item(), f"{switch}: logits moved by more than rounding"
        assert rel_upd <= 0.2, f"{switch}: update differs by {rel_upd:.2e}"


@pytest.mark.parametrize("arch,L", [("bert", 200), ("distilbert", 96)])
def test_bert_large_widths_and_sequences_over_128_tokens(arch, L):
    """dim 1024 / 16 heads / FFN 4096 (the LayerNorm kernels' widest row, N = 1024 / 3072 / 4096 GEMM tilings) and L = 200 (the
    non-persistent attention kernels: L > 128) through the whole step, weights at the HF init scale: logits against the oracle at
    the full-size bar (5e-3 of the logit scale), gradient direction, and a finite update."""
    cfg = EncoderConfig(arch=arch, vocab_size=1000, dim=1024, n_heads=16, hidden_dim=4096, n_layers=2, max_position_embeddings=256,
                        dropout=0.0, attention_dropout=0.0)
    model = selftest.build_tiny_model(cfg, std=0.02).cuda().train()
    batch = syn.nway_batch(4680, 2, 3, 12, L, vocab=cfg.vocab_size, ragged=True)
    tr = NwayTrainer(model, loss="margin_mse")
    _, logits = tr.forward_backward(batch)
    ref_logits, _, qp, pp = oracle_run(model, cfg, batch, "margin_mse")
    err = np.abs(logits.cpu().numpy() - ref_logits).max() / np.abs(ref_logits).max()
    assert err <= 5e-3, err
    got = torch.cat([p.grad.detach().reshape(-1) for _, p in model.passage_encoder.named_flat()]).double().cpu()
    want = torch.cat([pp[k].grad.reshape(-1) for k, _ in model.passage_encoder.named_flat()]).double()
    c = torch.nn.functional.cosine_similarity(got, want, dim=0).item()
    assert c >= 0.999 and abs((got.norm() / want.norm()).item() - 1.0) <= 2e-2, (c, (got.norm() / want.norm()).item())
    out = tr.train_step(batch)
    assert torch.isfinite(out[0]).item() and torch.isfinite(tr.flat_p).all().item()


@pytest.mark.parametrize("B,N,Lq,Lp,lens,loss", [(1, 1, 4, 8, None, "margin_mse"), (1, 2, 4, 8, None, "kl_div"), (2, 3, 1, 1, None, "margin_mse"),
                                                  (2, 3, 4, 8, [1] * 6, "margin_mse"), (2, 3, 4, 8, [1, 8, 3, 1, 8, 2], "kl_div")])
def test_degenerate_batch_shapes_train(B, N, Lq, Lp, lens, loss):
    """One query, one passage, one-token sequences, packed batches whose rows are single [CLS] tokens: two training steps with dropout
    run and leave finite weights; the first step's logits equal the oracle's."""
    cfg = small_cfg("distilbert", 2)
    model = selftest.build_tiny_model(cfg).cuda().train()
    batch = syn.nway_batch(1, B, N, Lq, Lp, vocab=cfg.vocab_size, ragged=False)
    if lens is not None:
        m = batch["nway_passages"]["attention_mask"].view(-1, Lp)
        for i, l in enumerate(lens):
            m[i, l:] = 0
        batch["nway_passages"]["lengths"] = torch.tensor(lens)
    tr = NwayTrainer(model, loss=loss)
    _, logits = tr.forward_backward(batch)
    ref = oracle_run(model, cfg, {k: v for k, v in batch.items()}, loss)[0]
    # (one-token sequences: every CLS vector is the same, the logits are small next to |q| |p| (~20) and carry its rounding: + 0.1)
    assert np.abs(logits.cpu().numpy() - ref).max() <= 3e-2 * np.abs(ref).max() + 0.1
    cfg.dropout = cfg.attention_dropout = 0.1
    m2 = selftest.build_tiny_model(cfg).cuda().train()
    tr2 = NwayTrainer(m2, loss=loss)
    for _ in range(2):
        out = tr2.train_step(batch)
    assert torch.isfinite(out[0]).item() and torch.isfinite(tr2.flat_p).all().item()


def test_packed_index_encode_matches_padded(monkeypatch):
    """The index path packs by default (retrieval_utils.batch_to_device takes the token counts from the host-side mask):
    get_embeddings_from_scratch on ragged batches against the same call with packing off."""
    from cldrd_amd.dataset import SyntheticSequenceDataset
    from cldrd_amd.retriever import retrieval_utils as RU
    cfg = small_cfg("distilbert", 3)
    model = selftest.build_tiny_model(cfg).cuda().eval()

    class Ragged(SyntheticSequenceDataset):
        def __getitem__(self, b):
            batch = super().__getitem__(b)
            m = batch["seq"]["attention_mask"]
            lens = 3 + syn.randint(900 + b, 0, m.shape[1] - 3, m.shape[0])
            mask = (np.arange(m.shape[1])[None, :] < lens[:, None]).astype(np.int64)
            batch["seq"] = {"input_ids": torch.from_numpy(np.where(mask == 1, batch["seq"]["input_ids"].numpy(), 0)), "attention_mask": torch.from_numpy(mask)}
            return batch

    ds = Ragged(700, 40, vocab=cfg.vocab_size, batch_size=256)
    monkeypatch.setenv("CLDRD_PACK", "1")
    e1, ids1 = RU.get_embeddings_from_scratch(model, ds.loader(), use_fp16=True, is_query=False)
    monkeypatch.setenv("CLDRD_PACK", "0")
    e0, ids0 = RU.get_embeddings_from_scratch(model, ds.loader(), use_fp16=True, is_query=False)
    assert ids1 == ids0 == list(range(700))
    print(f"packed index encode: identical rows {np.mean(np.all(e1 == e0, axis=1)):.2f}, max rel diff {np.abs(e1 - e0).max() / np.abs(e0).max():.2e}")
    assert np.abs(e1 - e0).max() <= 2e-3 * np.abs(e0).max()


def test_length_bucketed_index_encode_equals_the_plain_batches():
    """Round 6: the token-cache path of index_text batches rows by LENGTH (CachedSequenceDataset(bucket_window=...)) and
    get_embeddings_from_scratch puts them back in collection order.  Against the plain 512-consecutive-row batches of the same cache: ids in the
    same (file) order, and every passage's embedding the same - a passage's CLS vector does not depend on what it is batched with."""
    from cldrd_amd.dataset import CachedSequenceDataset, SequenceTokenCache
    from cldrd_amd.retriever import retrieval_utils as RU
    cfg = small_cfg("distilbert", 3)
    model = selftest.build_tiny_model(cfg).cuda().eval()
    rng = np.random.default_rng(11)
    n, L = 3000, 48
    lens = np.clip(np.round(rng.lognormal(2.6, 0.5, n)), 2, L).astype(np.int32)
    ids = np.zeros((n, L), dtype=np.uint16)
    for r in range(n):
        ids[r, :lens[r]] = rng.integers(5, cfg.vocab_size, lens[r])
        ids[r, 0] = 1
    cache = SequenceTokenCache(np.arange(n, dtype=np.int64) * 3 + 7, ids, lens, {"rows": n, "max_length": L})
    plain = CachedSequenceDataset(cache, 40, 2900, batch_size=512)
    e0, i0 = RU.get_embeddings_from_scratch(model, plain.loader(num_workers=0), True, False)
    buck = CachedSequenceDataset(cache, 40, 2900, batch_size=512, bucket_window=1024, token_budget=4096)
    assert len(buck) > len(plain)
    e1, i1 = RU.get_embeddings_from_scratch(model, buck.loader(num_workers=2), True, False)
    assert i1 == i0 == (np.arange(40, 2900) * 3 + 7).tolist()
    same = np.all(e1 == e0, axis=1).mean()
    print(f"length-bucketed index encode: identical rows {same:.3f}, max rel diff {np.abs(e1 - e0).max() / np.abs(e0).max():.2e}")
    # (packed and padded batches of different shapes: the same arithmetic per row; the Linear layers of a small chunk may take the small-M GEMM
    # kernel, whose K split sums in another order - hence a bar at fp32 rounding of the 16-bit pipeline, not bit equality)
    assert np.abs(e1 - e0).max() <= 2e-3 * np.abs(e0).max()


@pytest.mark.parametrize("arch", ["distilbert", "bert"])
def test_deferred_layernorm_parameter_gradients_equal_the_immediate_ones(arch, monkeypatch):
    """The deferred LayerNorm-parameter gradients (the gamma / beta and preceding-bias gradients of a tower are reduced by one grouped
    launch next to each weight-gradient group) against the immediate per-LayerNorm reductions (test hook `ln_defer = False`): the whole
    gradient buffer bit for bit outside the embedding tables (float atomics)."""
    cfg = small_cfg(arch=arch, layers=3)
    model = selftest.build_tiny_model(cfg).cuda().train()
    tr = NwayTrainer(model, loss="kl_div")
    batch = syn.nway_batch(4690, 3, 4, 8, 32, vocab=cfg.vocab_size, ragged=True)
    grads = {}
    for mode in ("0", "1"):
        for t in model.towers():
            t.ln_defer = mode == "1"
        tr.flat_g.fill_(7.0)
        tr.forward_backward(batch)
        torch.cuda.synchronize()
        grads[mode] = tr.flat_g.clone()
    emb = torch.zeros_like(tr.flat_g, dtype=torch.bool)
    for tower, toff in zip(model.towers(), model._tower_offsets):
        for n in ("embeddings.word_embeddings.weight", "embeddings.position_embeddings.weight"):
            off, shape = tower.layout.entries[n]
            emb[toff + off:toff + off + shape[0] * shape[1]] = True
    assert torch.equal(grads["0"][~emb], grads["1"][~emb])
    assert torch.allclose(grads["0"][emb], grads["1"][emb], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("all_neg,key", [(True, "logits_inbatch_all"), (False, "logits_inbatch_next")])
def test_in_batch_negative_logits_at_full_size_match_the_reference(all_neg, key):
    """models/nway_dual_encoder.py:30-44 at cfg1's full size (DistilBERT-6L, B = 4, N = 8, L = 128): the in-batch-negative score layouts
    ([B, B*N] with every other sample's passages; [B, 2N] with the next sample's) against what the REFERENCE produced for the same seeded
    weights and batch (tests/golden/full_distilbert_cfg1.npz), same bar as the N-way logits: 5e-3 of max|logit|; the trainer's step on them
    (lambda_mrr with the -0.5 label fill, nway_listwise_1.py:341-344) runs and matches the oracle's loss on those logits."""
    g = np.load(os.path.join(GOLDEN, "full_distilbert_cfg1.npz"))
    model = _full_size_model("distilbert", 6)
    model.in_batch_loss, model.all_in_batch_neg = True, all_neg
    B, N, Lq, Lp = int(g["B"]), int(g["N"]), int(g["Lq"]), int(g["Lp"])
    batch = syn.nway_batch(4680, B, N, Lq, Lp, ragged=True)
    tr = NwayTrainer(model, loss="lambda_mrr")
    loss_out, logits = tr.forward_backward(batch)
    ref = g[key]
    got = logits.cpu().numpy()
    assert got.shape == ref.shape == ((B, B * N) if all_neg else (B, 2 * N))
    err = np.abs(got - ref).max()
    print(f"in-batch ({key}): max|dlogit| {err:.4f} = {err / np.abs(ref).max():.2e} of max|logit| {np.abs(ref).max():.2f}")
    assert err <= 5e-3 * np.abs(ref).max()
    labels = np.concatenate([batch["labels"].numpy(), np.full((B, ref.shape[1] - N), -0.5, np.float32)], 1)
    assert loss_out[0].item() == pytest.approx(LR.lambda_mrr(got, labels)[0], rel=2e-5)
    assert torch.isfinite(tr.flat_g).all().item()
