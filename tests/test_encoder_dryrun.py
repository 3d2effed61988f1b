"""Host logic of the encoder / trainer without a GPU: every kernel launch is stubbed (`_lib.call` does nothing, the device check of the argument
validators is lifted), so a forward + backward walks the whole Python control flow - buffer formats, tape contents, the argument validation of
every hip_ops wrapper - in every mode: padded / packed, training / eval, fp16-operand FFN + QKV on / off, DistilBERT / BERT, CLS-only last layer.
Nothing is computed (the oracle and the GPU tests check values); this catches a wrong dtype, shape or missing buffer before a GPU box sees it."""
import numpy as np
import pytest
import torch

import cldrd_amd.synthetic as syn
from cldrd_amd import _lib, hip_ops as ops
from cldrd_amd.encoder import EncoderConfig, HipEncoder
from cldrd_amd.models import NwayDualEncoder


@pytest.fixture
def stubbed(monkeypatch):
    try:
        _lib.load()
    except Exception as e:
        pytest.skip(f"libcldrd_hip.so not built: {e}")
    calls = []

    def chk(t, dtype, name, dim=None):
        if not isinstance(t, torch.Tensor):
            raise TypeError(f"{name}: expected a tensor")
        if t.dtype != dtype:
            raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
        if dim is not None and t.dim() != dim:
            raise ValueError(f"{name}: expected {dim} dims, got {t.dim()}")
        if t.stride(-1) != 1:
            raise ValueError(f"{name}: last dim must be contiguous")
        return t
    monkeypatch.setattr(ops, "_chk", chk)
    monkeypatch.setattr(ops, "call", lambda name, *a: calls.append(name))
    monkeypatch.setattr(ops, "_stream", lambda: 0)
    monkeypatch.setattr(HipEncoder, "refresh_shadows", lambda self, need_transposed=True, cast=True, cast16=None, h_stale=False: _cpu_shadows(self))
    monkeypatch.setattr(HipEncoder, "_shadows_ok", lambda self, need_t, need_h=True: self.flat_h is not None)
    return calls


def _cpu_shadows(enc):
    enc.flat_h = torch.zeros(enc.layout.total, dtype=torch.bfloat16)
    enc.flat_h16 = torch.zeros(enc.layout.total, dtype=torch.float16)
    enc.flat_t = torch.zeros(enc.layout.t_total, dtype=torch.float16 if enc.amp16 else torch.bfloat16)
    enc._t_fresh = True
    enc._shadow_version = enc.flat_p._version


def cfg_of(arch, layers):
    return EncoderConfig(arch=arch, vocab_size=512, dim=128, n_heads=2, hidden_dim=256, n_layers=layers, max_position_embeddings=64,
                         dropout=0.1, attention_dropout=0.1)


@pytest.mark.parametrize("arch,layers", [("distilbert", 3), ("bert", 2), ("distilbert", 1), ("bert", 7)])
@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("amp,stream16", [("fp16", True), ("fp16", False), ("bf16", True)])
def test_forward_backward_walks_every_mode(stubbed, monkeypatch, arch, layers, packed, amp, stream16):
    """CLDRD_AMP = fp16 (all-fp16 training pass; evaluation with fp16 FFN / out-projection operands, fp16 QKV operands on towers deeper than 6
    layers: the 7-layer case) and bf16 (every operand bf16), the fp16 mode also with its fp32 gradient stream (test hook)."""
    monkeypatch.setenv("CLDRD_AMP", amp)
    enc = HipEncoder(cfg_of(arch, layers), seed=1)
    enc._grad_stream16 = stream16
    assert enc.amp16 == (amp == "fp16") and enc.grad_stream16 == (amp == "fp16" and stream16) and enc.needs_h16 == (amp == "fp16")
    t16 = torch.float16 if enc.amp16 else torch.bfloat16
    M, L = 6, 24
    lens = np.array([24, 3, 10, 17, 5, 8])
    ids = torch.randint(3, 500, (M, L))
    mask = (torch.arange(L)[None, :] < torch.from_numpy(lens)[:, None]).long()
    lengths = lens.tolist() if packed else None
    for hp in (False, True):
        enc.hp_forward = hp
        cls = enc.encode(ids, mask, train=False, save=False, lengths=lengths)
        assert cls.shape == (M, 128) and cls.dtype == torch.float32
        cls, tape = enc.encode(ids, mask, train=True, save=True, lengths=lengths)
        pk = packed                                 # one pass per mode for both towers: a training pass packs whenever it is given lengths
        assert (tape.pack is not None) == pk and tape.T == (int(lens.sum()) if pk else M * L)
        for a in tape.layers:                       # what the backward's MFMAs read: one 16-bit format per mode
            for k in ("x_in", "ctx", "x1", "h", "pre"):
                assert a[k] is not None and a[k].dtype == t16, (k, a[k].dtype)
            assert (a.get("qkv") if "qkv" in a else a["kv"]).dtype == t16
        hooks = []
        enc.backward_from_cls(tape, torch.zeros(M, 128), after_layer=hooks.append, accumulate=False)
        assert sorted(hooks) == list(range(-1, layers))        # every bucket reported once (the embedding block before the last weight-gradient group)
    kinds = set(stubbed)
    assert "cldrd_gemm_nt16_ws" in kinds and "cldrd_wgrad_group" in kinds and "cldrd_embed_ln_bwd" in kinds
    assert ("cldrd_attention_cls_fwd_varlen" in kinds) == packed and ("cldrd_attention_cls_bwd_varlen" in kinds) == packed
    assert ("cldrd_attention_fwd_varlen" in kinds) == (packed and layers > 1) and ("cldrd_attention_bwd_varlen" in kinds) == (packed and layers > 1)
    assert "cldrd_unpack_rows16" not in kinds                      # round 6: attention reads the packed rows through cu, no row moves


@pytest.mark.parametrize("lens,want", [([200, 3, 10, 130, 5, 128], 2), ([100, 3, 10, 128, 5, 64], 1), ([200, 150, 129, 130, 131, 199], 0)])
def test_packed_batches_above_128_tokens_split_their_attention_launches_by_length(stubbed, monkeypatch, lens, want):
    """encoder._Pack at L > 128: the sequences of at most 128 tokens go through the L <= 128 kernels (cldrd_attention_*_varlen_list with tile 128),
    the others through a second launch at tile L; all short: one listed launch; all long: the plain packed launch."""
    monkeypatch.setenv("CLDRD_AMP", "fp16")
    cfg = EncoderConfig(arch="distilbert", vocab_size=512, dim=128, n_heads=2, hidden_dim=256, n_layers=2, max_position_embeddings=256,
                        dropout=0.1, attention_dropout=0.1)
    enc = HipEncoder(cfg, seed=1)
    lens = lens * 4                                   # 24 sequences: enough tokens for the encoder to pack (would_pack: >= 1024)
    M, L = len(lens), 200
    ids = torch.randint(3, 500, (M, L))
    mask = (torch.arange(L)[None, :] < torch.tensor(lens)[:, None]).long()
    n0 = len(stubbed)
    cls, tape = enc.encode(ids, mask, train=True, save=True, lengths=lens)
    assert tape.pack is not None and [t for _, t in tape.pack.groups] == ([128, L] if want == 2 else [128] if want == 1 else [0])
    if want:
        assert sorted(int(i) for sl, _ in tape.pack.groups for i in sl.tolist()) == list(range(M))
    enc.backward_from_cls(tape, torch.randn(M, cfg.dim))
    calls = stubbed[n0:]
    fwd_l, bwd_l = calls.count("cldrd_attention_fwd_varlen_list"), calls.count("cldrd_attention_bwd_varlen_list")
    assert fwd_l == bwd_l == want * 1 and calls.count("cldrd_attention_fwd_varlen") == calls.count("cldrd_attention_bwd_varlen") == (0 if want else 1)


def test_trainer_step_walks_with_stubbed_kernels(stubbed, monkeypatch):
    """forward_backward + optimizer launches of NwayTrainer (padded and packed batch) with stubbed kernels; `lengths` travel from the
    host-side mask (trainer.batch_to_device) to the passage tower."""
    from cldrd_amd.trainer import nway_listwise as TL
    monkeypatch.setattr(TL.NwayTrainer, "_require_gpu", staticmethod(lambda dev: None))
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: _FakeStream())
    monkeypatch.setattr(torch.cuda, "stream", lambda s: _Null())
    monkeypatch.setattr(torch.cuda, "Stream", lambda *a, **k: _FakeStream())
    monkeypatch.setenv("CLDRD_GRAPH", "0")
    model = NwayDualEncoder(cfg_of("distilbert", 2), share_weights=False)
    tr = TL.NwayTrainer(model, loss="kl_div")
    tr.q_stream = None
    batch = syn.nway_batch(5, 2, 3, 6, 20, vocab=512, ragged=True, label_kind="teacher")
    moved = TL.batch_to_device(batch, torch.device("cpu"))
    assert moved["nway_passages"]["lengths"].tolist() == batch["nway_passages"]["attention_mask"].sum(-1).reshape(-1).tolist()
    n0 = len(stubbed)
    tr.train_step(moved)
    assert tr.global_step == 1 and "cldrd_adamw_step_h16" in stubbed[n0:] and "cldrd_loss_fwd_bwd" in stubbed[n0:]


def test_window_scheduled_step_walks_and_interleaves_the_two_towers(stubbed, monkeypatch):
    """Round 6: the query tower is enqueued in slices released at the passage tower's LayerNorm / attention launches (encoder.Stepper,
    NwayTrainer._window).  With stubbed kernels and fake streams: the launch sequence of a step must contain every launch of the free-running
    schedule exactly once (same multiset), the query tower's launches must be INTERLEAVED with the passage tower's (not all in front), and
    forward and backward must each have drained their stepper."""
    from cldrd_amd.trainer import nway_listwise as TL
    monkeypatch.setattr(TL.NwayTrainer, "_require_gpu", staticmethod(lambda dev: None))
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: _MAIN)
    monkeypatch.setattr(torch.cuda, "stream", lambda s_: _Null())
    monkeypatch.setattr(torch.cuda, "Stream", lambda *a, **k: _FakeStream())
    monkeypatch.setattr(torch.cuda, "Event", lambda *a, **k: _FakeEvent())
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    monkeypatch.setattr(torch.Tensor, "record_stream", lambda self, s_: None)
    monkeypatch.setenv("CLDRD_GRAPH", "0")
    seqs = {}
    for windows in (False, True):
        torch.manual_seed(0)
        model = NwayDualEncoder(cfg_of("distilbert", 3), share_weights=False)
        tr = TL.NwayTrainer(model, loss="kl_div")
        tr.q_stream, tr.l_stream = _FakeStream(), _FakeStream()
        tr.window_schedule = windows
        batch = syn.nway_batch(5, 2, 3, 6, 20, vocab=512, ragged=False, label_kind="teacher")
        n0 = len(stubbed)
        tr.train_step(batch)
        seqs[windows] = list(stubbed[n0:])
        assert tr.global_step == 1
    assert sorted(seqs[True]) == sorted(seqs[False]) and seqs[True] != seqs[False]
    # free-running: the query tower's whole forward (its embedding kernel first) is enqueued before the passage tower's first launch;
    # windowed: the passage tower's embedding kernel comes first and the two towers' attention launches alternate
    first_attn = [i for i, n in enumerate(seqs[True]) if n.startswith("cldrd_attention_fwd")]
    assert len(first_attn) >= 4
    emb = [i for i, n in enumerate(seqs[True]) if n == "cldrd_embed_ln_fwd"]
    assert len(emb) == 2 and emb[1] - emb[0] <= 3           # both embedding launches at the very start of the windowed forward



class _FakeEvent:
    def record(self, stream=None):
        pass


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _FakeStream:
    cuda_stream = 0

    def wait_stream(self, s):
        pass

    def wait_event(self, e):
        pass


def test_would_pack_rules():
    """Packing is chosen from the host-side token counts: needs a mask and lengths, at least 8 % padding, and never moves the Linear layers across the
    M = 1024 boundary between the large-M and the small-M GEMM kernels (their summation orders differ: packed == padded would stop being bit-exact)."""
    enc = HipEncoder(cfg_of("distilbert", 2), seed=1)
    M, L = 64, 32                                     # 2048 padded rows
    full = [L] * M
    assert not enc.would_pack(full, M, L)             # no padding at all
    assert not enc.would_pack(None, M, L)
    assert enc.would_pack([20] * M, M, L)             # 1280 tokens: both sides of the run on the large-M kernel
    assert not enc.would_pack([20] * M, M, L, has_mask=False)
    assert not enc.would_pack([10] * M, M, L)         # 640 tokens < 1024 <= 2048 padded rows: stays padded
    assert enc.would_pack([10] * 16, 16, L)           # 160 of 512: both small-M
    assert not enc.would_pack([31] * M, M, L)         # 3 % padding: not worth the row moves
    assert not enc.would_pack([20] * (M - 1) + [0], M, L)         # an empty row: the packed kernels take 1 .. L rows per sequence
    assert not enc.would_pack([20] * (M - 1) + [L + 1], M, L) and not enc.would_pack([20] * (M - 1), M, L)      # counts of another batch shape


_MAIN = _FakeStream()
