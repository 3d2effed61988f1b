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


@pytest.mark.parametrize("arch,layers", [("distilbert", 3), ("bert", 2), ("distilbert", 1)])
@pytest.mark.parametrize("ffn16,qkv16,out16", [("1", "1", "1"), ("1", "0", "1"), ("1", "auto", "0"), ("0", "0", "1")])
@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("gs,amp", [("fp32", "fp16"), ("fp32", "bf16"), ("bf16", "fp16")])
def test_forward_backward_walks_every_mode(stubbed, monkeypatch, arch, layers, ffn16, qkv16, out16, packed, gs, amp):
    monkeypatch.setenv("CLDRD_AMP", amp)
    monkeypatch.setenv("CLDRD_FFN_FP16", ffn16)
    monkeypatch.setenv("CLDRD_QKV_FP16", qkv16)
    monkeypatch.setenv("CLDRD_OUT_FP16", out16)
    monkeypatch.setenv("CLDRD_GRAD_STREAM", gs)
    enc = HipEncoder(cfg_of(arch, layers), seed=1)
    assert enc.amp16 == (amp == "fp16" and gs == "fp32")        # the all-fp16 training mode needs the fp32 gradient stream
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
        # the dual-pass (fp16 + bf16 tape) forward of the query tower is never packed; the all-fp16 mode has one pass for both towers
        pk = packed and (enc.amp16 or not hp)
        assert (tape.pack is not None) == pk and tape.T == (int(lens.sum()) if pk else M * L)
        for a in tape.layers:                       # what the backward's MFMAs read: one 16-bit format per mode
            for k in ("x_in", "ctx", "x1", "h", "pre"):
                assert a[k] is not None and a[k].dtype == t16, (k, a[k].dtype)
            assert (a.get("qkv") if "qkv" in a else a["kv"]).dtype == t16
        hooks = []
        enc.backward_from_cls(tape, torch.zeros(M, 128), after_layer=hooks.append, accumulate=False)
        assert sorted(hooks) == list(range(-1, layers))        # every bucket reported once (the embedding block before the last weight-gradient group)
    kinds = set(stubbed)
    assert "cldrd_gemm_nt_bf16_ws" in kinds and "cldrd_wgrad_group" in kinds and "cldrd_embed_ln_bwd" in kinds
    assert ("cldrd_unpack_rows16" in kinds) == packed


def test_trainer_step_walks_with_stubbed_kernels(stubbed, monkeypatch):
    """forward_backward + optimizer launches of NwayTrainer (padded and packed batch) with stubbed kernels; `lengths` travel from the
    host-side mask (trainer.batch_to_device) to the passage tower."""
    from cldrd_amd.trainer import nway_listwise as TL
    monkeypatch.setattr(TL.NwayTrainer, "_require_gpu", staticmethod(lambda dev: None))
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: _FakeStream())
    monkeypatch.setattr(torch.cuda, "stream", lambda s: _Null())
    monkeypatch.setattr(torch.cuda, "Stream", lambda *a, **k: _FakeStream())
    monkeypatch.setenv("CLDRD_GRAPH", "0")
    monkeypatch.setenv("CLDRD_Q_SIDE", "0")
    model = NwayDualEncoder(cfg_of("distilbert", 2), share_weights=False)
    tr = TL.NwayTrainer(model, loss="kl_div")
    tr.q_stream = None
    batch = syn.nway_batch(5, 2, 3, 6, 20, vocab=512, ragged=True, label_kind="teacher")
    moved = TL.batch_to_device(batch, torch.device("cpu"))
    assert moved["nway_passages"]["lengths"].tolist() == batch["nway_passages"]["attention_mask"].sum(-1).reshape(-1).tolist()
    n0 = len(stubbed)
    tr.train_step(moved)
    assert tr.global_step == 1 and "cldrd_adamw_step_h16" in stubbed[n0:] and "cldrd_loss_fwd_bwd" in stubbed[n0:]


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
