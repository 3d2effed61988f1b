"""The data-parallel training path executed for real: two ranks, one process each, BOTH on the one GPU of the
test box, torch.distributed over gloo (it accepts device tensors) - so the code under test is exactly what runs over RCCL on an
8-GPU node except for the transport: DDP-style constructor broadcast (reference nway_listwise_1.py:250-255), per-layer gradient
buckets all-reduced from the backward hooks on the side stream while earlier layers are still in backward, 1/world folded into
dlogits, write-once gradients.  The reduced gradient must equal the single-process gradient of the concatenated batch."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out, share):
    import numpy as np
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cldrd_amd.synthetic as syn
    import selftest
    from cldrd_amd.encoder import EncoderConfig
    from cldrd_amd.trainer import NwayTrainer
    from cldrd_amd.trainer.nway_listwise import common_steps_per_epoch

    cfg = EncoderConfig(arch="distilbert", vocab_size=512, dim=128, n_heads=2, hidden_dim=256, n_layers=3,
                        max_position_embeddings=64, dropout=0.0, attention_dropout=0.0)
    # different initial weights per rank: the constructor broadcast must make rank 0's win
    model = selftest.build_tiny_model(cfg, share_weights=share, seed=3 + 10 * rank).cuda()
    model.train()
    tr = NwayTrainer(model, loss="margin_mse", learning_rate=1e-3, warmup_steps=0, total_steps=10)
    assert tr.distributed and tr.world == world
    ref_model = selftest.build_tiny_model(cfg, share_weights=share, seed=3).cuda()
    ref_model.fuse_flat()
    assert torch.equal(tr.flat_p, ref_model._flat_p), "rank 0's parameters were not broadcast"
    B = 2
    full = syn.nway_batch(4680, B * world, 5, 10, 40, vocab=cfg.vocab_size, ragged=True)

    def rows(t, lo, hi):
        return t[lo:hi]
    mine = {k: ({kk: rows(vv, rank * B, (rank + 1) * B) for kk, vv in v.items()} if isinstance(v, dict) else rows(v, rank * B, (rank + 1) * B))
            for k, v in full.items()}
    tr.flat_g.fill_(77.0)                      # garbage: every gradient must be written (and reduced) exactly once
    loss_out, _ = tr.forward_backward(mine)
    torch.cuda.synchronize()
    reduced = tr.flat_g.clone()
    # every rank holds the same reduced gradient
    gathered = [torch.empty_like(reduced) for _ in range(world)]
    dist.all_gather(gathered, reduced)
    assert all(torch.equal(gathered[0], g) for g in gathered), "ranks disagree on the reduced gradient"
    # equal step counts: rank r would have 5 + r batches
    assert common_steps_per_epoch(5 + rank, True, torch.device("cuda", 0)) == 5
    if rank == 0:
        tr.distributed, tr.world = False, 1            # the same trainer, single process, the concatenated batch
        tr.flat_g.fill_(-5.0)
        tr.forward_backward(full)
        torch.cuda.synchronize()
        single = tr.flat_g
        scale = single.abs().max().item()
        err = (reduced - single).abs().max().item()
        # same kernels on the same rows; only the token-split of the weight-gradient reduction and the embedding atomics differ
        assert err <= 2e-3 * scale, f"DDP gradient differs from the single-process gradient: {err:.3e} vs scale {scale:.3e}"
        # one optimizer step on the reduced gradients keeps the ranks' weights identical
        open(out, "w").write(f"ok {err / scale:.2e}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("share", [False, True])
def test_ddp_bucket_hooks_world2_on_one_gpu(tmp_path, share):
    import torch.multiprocessing as mp
    out = str(tmp_path / "ok")
    mp.spawn(_worker, args=(2, _free_port(), out, share), nprocs=2, join=True)
    assert open(out).read().startswith("ok")


@pytest.mark.parametrize("world", [4, 8])
def test_ddp_bucket_hooks_world4_and_world8_on_one_gpu(tmp_path, world):
    """The same at the world sizes of the scaling table (round-3 review): 4 and 8 rank processes share the one GPU; bucket plan, constructor
    broadcast, 1 / world in dlogits, MIN step count and the reduced gradient against the single-process gradient of the concatenated
    batch (8 ranks x 2 queries).  Still gloo: what crosses xGMI is not tested here."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "ok")
    mp.spawn(_worker, args=(world, _free_port(), out, False), nprocs=world, join=True)
    assert open(out).read().startswith("ok")


def _rccl_worker(rank, port, out):
    """ONE rank over the real backend: `init_process_group("nccl")` is RCCL on ROCm.  With CLDRD_FORCE_DDP=1 the trainer takes its
    data-parallel path although world_size is 1, so everything that path does with ProcessGroupNCCL runs on the GPU for real: the
    constructor broadcast, the per-bucket `all_reduce(async_op=True)` issued from the backward hooks on the communication stream behind an
    event of the compute stream, the Work handles waited for in `_wait_pending`, MIN over ranks of the step count.  An all-reduce over
    one rank is the identity, so gradients and updated weights must equal the plain single-process trainer's."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", CLDRD_FORCE_DDP="1",
                      HSA_ENABLE_IPC_MODE_LEGACY="0", GPU_MAX_HW_QUEUES="8")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    assert dist.get_backend() == "nccl"
    import cldrd_amd.synthetic as syn
    import selftest
    from cldrd_amd.encoder import EncoderConfig
    from cldrd_amd.trainer import NwayTrainer
    from cldrd_amd.trainer.nway_listwise import common_steps_per_epoch

    cfg = EncoderConfig(arch="distilbert", vocab_size=512, dim=128, n_heads=2, hidden_dim=256, n_layers=3,
                        max_position_embeddings=64, dropout=0.0, attention_dropout=0.0)
    batch = syn.nway_batch(4680, 4, 5, 10, 40, vocab=cfg.vocab_size, ragged=True)
    model = selftest.build_tiny_model(cfg, seed=3).cuda()
    model.train()
    tr = NwayTrainer(model, loss="margin_mse", learning_rate=1e-3, warmup_steps=0, total_steps=10)
    assert tr.distributed and tr.world == 1 and tr.comm_stream is not None
    assert common_steps_per_epoch(7, True, torch.device("cuda", 0)) == 7          # all_reduce(MIN) of a device tensor over RCCL
    tr.flat_g.fill_(77.0)
    tr.forward_backward(batch)
    assert not tr._pending, "bucket all-reduces left unwaited"
    torch.cuda.synchronize()
    g_ddp = tr.flat_g.clone()
    os.environ["CLDRD_FORCE_DDP"] = "0"
    plain_model = selftest.build_tiny_model(cfg, seed=3).cuda()
    plain_model.train()
    plain = NwayTrainer(plain_model, loss="margin_mse", learning_rate=1e-3, warmup_steps=0, total_steps=10)
    assert not plain.distributed
    plain.forward_backward(batch)
    torch.cuda.synchronize()
    scale = plain.flat_g.abs().max().item()
    err = (g_ddp - plain.flat_g).abs().max().item()
    # same kernels; the weight-gradient groups are flushed every 2 layers instead of once (other token splits) and the embedding atomics race
    assert err <= 2e-3 * scale, f"gradient through the RCCL path differs: {err:.3e} vs scale {scale:.3e}"
    p0 = plain.flat_p.clone()
    for _ in range(3):                                     # whole steps: hooks + clip + AdamW (the first three of a batch shape are eager)
        tr.train_step(batch)
        plain.train_step(batch)
    torch.cuda.synchronize()
    # the clip norm is the norm of the reduced gradient
    assert abs(tr.clip[0].item() - tr.flat_g.double().norm().item()) <= 1e-5 * tr.clip[0].item()
    assert abs(tr.clip[0].item() - plain.clip[0].item()) <= 2e-3 * plain.clip[0].item()
    assert torch.isfinite(tr.flat_p).all().item()
    # AdamW turns a last-bit gradient difference on a near-zero gradient (k_lin.bias: mathematically zero) into +-lr, so single
    # weights may differ by a few lr; the UPDATE as a whole must agree
    upd, upd_ref = (tr.flat_p - p0).double(), (plain.flat_p - p0).double()
    dp = ((upd - upd_ref).norm() / upd_ref.norm()).item()
    assert upd_ref.norm().item() > 0 and dp <= 0.05, f"update after 3 steps differs: relative {dp:.3e}"
    # round 4: from the 4th step of a batch shape on, a data-parallel rank over ProcessGroupNCCL replays the step as a HIP graph with the
    # bucket all-reduces captured inside (gloo ranks stay eager); the plain trainer replays its own graph: the two keep agreeing
    for _ in range(4):
        l_ddp = tr.train_step(batch).clone()
        l_plain = plain.train_step(batch).clone()
    torch.cuda.synchronize()
    assert any(e["graph"] is not None for e in tr._graphs.values()), "the data-parallel step was not captured"
    assert not getattr(tr, "_graph_broken", False)
    assert torch.allclose(l_ddp, l_plain, rtol=2e-2, atol=1e-6), (l_ddp, l_plain)      # lr = 1e-3 on a tiny model amplifies last-bit differences
    upd, upd_ref = (tr.flat_p - p0).double(), (plain.flat_p - p0).double()
    dp7 = ((upd - upd_ref).norm() / upd_ref.norm()).item()
    assert dp7 <= 0.05, f"update after 7 steps (4 of them replayed with RCCL inside the graph) differs: relative {dp7:.3e}"
    open(out, "w").write(f"ok {err / scale:.2e} {dp:.2e} {dp7:.2e}")
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_path_over_rccl_with_one_rank(tmp_path):
    """What one GPU can prove about the RCCL path (tests above: two ranks over gloo): ProcessGroupNCCL itself, world_size 1."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "ok")
    mp.spawn(_rccl_worker, args=(_free_port(), out), nprocs=1, join=True)
    assert open(out).read().startswith("ok")


def _rccl_replay_worker(rank, port, out):
    """200 replays of the captured data-parallel step (bucket all-reduces over ProcessGroupNCCL inside the graph) against the EAGER
    data-parallel step, each pair from identical state: loss, logits and every gradient outside the embedding tables (float atomics) bit for
    bit.  One rank: what is proven is that capture + replay of the collectives and their event edges reproduce the eager ordering - a
    replay that ran an all-reduce before its bucket was complete, or the norm before an all-reduce, would differ."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", CLDRD_FORCE_DDP="1",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    import cldrd_amd.synthetic as syn
    import selftest
    from cldrd_amd.encoder import EncoderConfig
    from cldrd_amd.trainer import NwayTrainer

    cfg = EncoderConfig(arch="distilbert", vocab_size=512, dim=128, n_heads=2, hidden_dim=256, n_layers=3,
                        max_position_embeddings=64, dropout=0.1, attention_dropout=0.1)
    batches = [syn.nway_batch(4680 + i, 4, 5, 10, 40, vocab=cfg.vocab_size, ragged=False, label_kind="teacher") for i in range(4)]
    batches = [{k: ({kk: vv.cuda() for kk, vv in v.items()} if isinstance(v, dict) else v.cuda()) for k, v in b.items()} for b in batches]

    def make():
        m = selftest.build_tiny_model(cfg, seed=3).cuda()
        m.train()
        return NwayTrainer(m, loss="kl_div", learning_rate=1e-4, warmup_steps=0, total_steps=10 ** 6)
    G, E = make(), make()
    assert G.distributed and E.distributed

    def step(tr, b, graph):
        os.environ["CLDRD_DDP_GRAPH"] = "1" if graph else "0"
        return tr.train_step(b)
    for i in range(4):
        step(G, batches[0], True)
    assert any(e["graph"] is not None for e in G._graphs.values()) and not getattr(G, "_graph_broken", False)
    keep = torch.ones(G.flat_g.numel(), dtype=torch.bool, device="cuda")
    for tower, toff in zip(G.model.towers(), G.model._tower_offsets):
        for name in tower.layout.order:
            if name.startswith("embeddings.") and name.endswith("_embeddings.weight"):
                off, shape = tower.layout.entries[name]
                n = 1
                for s_ in shape:
                    n *= s_
                keep[toff + off:toff + off + n] = False
    worst = 0
    for i in range(200):
        # identical state: parameters, moments, step counters, dropout counters, loss-scale block
        E.flat_p.copy_(G.flat_p), E.m.copy_(G.m), E.v.copy_(G.v)
        E.global_step, E.adam_step = G.global_step, G.adam_step
        if G._scale_state is not None:
            E._scale_state.copy_(G._scale_state)
        for tg, te in zip(G.model.towers(), E.model.towers()):
            te.step_seed = tg.step_seed
            te.refresh_shadows(need_transposed=True)
        b = batches[i % len(batches)]
        lg = step(G, b, True).clone()
        le = step(E, b, False).clone()
        torch.cuda.synchronize()
        assert not any(e["graph"] is not None for e in getattr(E, "_graphs", {}).values())
        assert torch.equal(lg, le), (i, lg, le)
        assert torch.equal(G.last_logits, E.last_logits), i
        same = torch.equal(G.flat_g[keep], E.flat_g[keep])
        if not same:
            worst = max(worst, int((G.flat_g[keep] != E.flat_g[keep]).sum().item()))
        assert same, f"replay {i}: {worst} gradient elements outside the embedding tables differ from the eager data-parallel step"
    open(out, "w").write("ok 200")
    dist.barrier()
    dist.destroy_process_group()


def test_captured_ddp_step_replays_bit_for_bit_200_times(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "ok")
    mp.spawn(_rccl_replay_worker, args=(_free_port(), out), nprocs=1, join=True)
    assert open(out).read().startswith("ok 200")


def _rccl_agreement_worker(rank, port, out):
    """the one-off capture agreement of data-parallel ranks happens at train_step CALL number warm + 1 on every rank,
    whatever that rank's batches looked like - the count advances in front of the packed / padded gate.  A rank whose batch at that call is
    packed (its own token fill decides that), or that saw packed batches before, votes "no"; nobody reaches the all-reduce at another call
    or never (which would pair it with another rank's bucket all-reduces: a hang).  One rank over ProcessGroupNCCL: what is checked is WHEN
    the agreement collective is issued and with which vote."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", CLDRD_FORCE_DDP="1", CLDRD_DDP_GRAPH="1",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    import cldrd_amd.synthetic as syn
    import selftest
    from cldrd_amd.encoder import EncoderConfig
    from cldrd_amd.trainer import NwayTrainer
    from cldrd_amd.trainer.nway_listwise import batch_to_device

    cfg = EncoderConfig(arch="distilbert", vocab_size=512, dim=128, n_heads=2, hidden_dim=256, n_layers=2,
                        max_position_embeddings=64, dropout=0.0, attention_dropout=0.0)

    def batch(seed, packed):
        b = syn.nway_batch(seed, 4, 16, 10, 64, vocab=cfg.vocab_size, ragged=False, label_kind="teacher")
        if packed:       # ~45 % token fill: the encoder packs it (would_pack), so the step cannot be a graph replay
            lens = 20 + (torch.arange(64) % 20)
            m = (torch.arange(64)[None, :] < lens[:, None]).to(torch.int64).view(4, 16, 64)
            b["nway_passages"]["attention_mask"] = m
            b["nway_passages"]["input_ids"] = b["nway_passages"]["input_ids"] * m
        return batch_to_device(b, torch.device("cuda", 0))

    def scenario(pattern):
        m = selftest.build_tiny_model(cfg, seed=3).cuda()
        m.train()
        tr = NwayTrainer(m, loss="kl_div", learning_rate=1e-4, warmup_steps=0, total_steps=1000)
        votes, real = [], tr._agree_on_capture

        def spy(ok):
            votes.append((tr._ddp_steps, bool(ok)))
            return real(ok)
        tr._agree_on_capture = spy
        for i, packed in enumerate(pattern):
            bt = batch(100 + i, packed)
            if packed:
                assert "lengths" in bt["nway_passages"]
            tr.train_step(bt)
        torch.cuda.synchronize()
        return tr, votes
    W = NwayTrainer._DDP_WARM
    # every call padded: captured at call W + 1, replayed afterwards
    tr, votes = scenario([False] * (W + 3))
    assert votes == [(W + 1, True)], votes
    assert any(e["graph"] is not None for e in tr._graphs.values()) and not getattr(tr, "_graph_broken", False)
    # the batch of call W + 1 is packed: this rank still votes at call W + 1 ("no"), and never again
    tr, votes = scenario([False] * W + [True, False, False, True])
    assert votes == [(W + 1, False)], votes
    assert tr._graph_broken and not any(e["graph"] is not None for e in getattr(tr, "_graphs", {}).values())
    # packed batches BEFORE the agreement step: the padded batch of call W + 1 has not been seen `warm` times -> "no", at call W + 1
    tr, votes = scenario([True, True, False, False, False, False])
    assert votes == [(W + 1, False)], votes
    assert tr._graph_broken
    # always packed: the vote still happens, at the same call
    tr, votes = scenario([True] * (W + 2))
    assert votes == [(W + 1, False)], votes
    open(out, "w").write("ok agreement")
    dist.barrier()
    dist.destroy_process_group()


def test_capture_agreement_happens_at_a_fixed_call_on_every_rank(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "ok")
    mp.spawn(_rccl_agreement_worker, args=(_free_port(), out), nprocs=1, join=True)
    assert open(out).read().startswith("ok agreement")
