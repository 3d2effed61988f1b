"""Checkpoint / optimizer-state interop with the reference trainer (trainer/multistep-curriculum/nway_listwise_1.py:256-266,
:300-304, :418-426): the reference builds ``AdamW(optimizer_grouped_parameters)`` over ``model.named_parameters()`` of an
``NwayDualEncoder`` whose towers are HF ``AutoModel``s, saves ``optimizer.state_dict()`` and loads it back on resume.
The CPU tests here pin the pieces that do not need a GPU: HF's parameter order, the group / index mapping, the hub-name
policy of ``from_pretrained``, the token-cache metadata, the equal-steps rule.  The GPU round trip is in test_gpu_model.py."""
import os

import numpy as np
import pytest
import torch

from cldrd_amd.encoder import EncoderConfig, HipEncoder, hf_parameter_order
from cldrd_amd.models import NwayDualEncoder
from cldrd_amd.trainer import nway_listwise as T


def _hf_model(arch, cfg: EncoderConfig):
    from transformers import BertConfig, BertModel, DistilBertConfig, DistilBertModel
    if arch == "distilbert":
        return DistilBertModel(DistilBertConfig(vocab_size=cfg.vocab_size, dim=cfg.dim, n_heads=cfg.n_heads, hidden_dim=cfg.hidden_dim,
                                                n_layers=cfg.n_layers, max_position_embeddings=cfg.max_position_embeddings))
    return BertModel(BertConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.dim, num_attention_heads=cfg.n_heads,
                                intermediate_size=cfg.hidden_dim, num_hidden_layers=cfg.n_layers,
                                max_position_embeddings=cfg.max_position_embeddings))


def _cfg(arch):
    return EncoderConfig(arch=arch, vocab_size=64, dim=128, n_heads=2, hidden_dim=256, n_layers=2, max_position_embeddings=16)


@pytest.mark.parametrize("arch", ["distilbert", "bert"])
def test_hf_parameter_order_matches_installed_transformers(arch):
    cfg = _cfg(arch)
    hf = _hf_model(arch, cfg)
    assert [n for n, _ in hf.named_parameters()] == hf_parameter_order(cfg, with_pooler=True)
    own = HipEncoder(cfg)
    assert sorted(n for n, _ in own.named_parameters()) == sorted(hf_parameter_order(cfg, with_pooler=False))


@pytest.mark.parametrize("arch,share", [("distilbert", False), ("bert", False), ("distilbert", True)])
def test_optimizer_groups_follow_the_reference_construction(arch, share):
    """Build the optimizer exactly as nway_listwise_1.py:258-263 does over an HF-tower model and compare index -> name."""
    cfg = _cfg(arch)

    class Ref(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.query_encoder = _hf_model(arch, cfg)
            self.passage_encoder = self.query_encoder if share else _hf_model(arch, cfg)

    ref = Ref()
    no_decay = ["bias", "LayerNorm.weight"]
    groups = [[n for n, p in ref.named_parameters() if not any(nd in n for nd in no_decay)],
              [n for n, p in ref.named_parameters() if any(nd in n for nd in no_decay)]]
    model = NwayDualEncoder(cfg, share_weights=share)
    ours = T.optimizer_param_groups(model)
    assert [[e[0] for e in g] for g in ours] == groups
    for g in ours:
        for name, ti, hf_name in g:
            assert (ti is None) == ("pooler" in name)


def test_from_pretrained_refuses_hub_names_without_opt_in(monkeypatch):
    monkeypatch.delenv("CLDRD_ALLOW_RANDOM_INIT", raising=False)
    monkeypatch.setenv("HF_HUB_OFFLINE", "1")
    with pytest.raises(FileNotFoundError):
        HipEncoder.from_pretrained("sebastian-hofstaetter/distilbert-dot-tas_b-b256-msmarco")
    with pytest.raises(FileNotFoundError):
        HipEncoder.from_pretrained("no-such-org/no-such-model")
    with pytest.warns(UserWarning):
        enc = HipEncoder.from_pretrained("distilbert-base-uncased", allow_random_init=True)
    assert enc.cfg.n_layers == 6 and enc.cfg.dim == 768


def test_parameters_share_the_flat_version_counter():
    """optimizer.step() / load_state_dict through the nn.Parameters must be visible on flat_p._version (what the bf16 weight
    shadows are keyed on), also after the parameters were re-homed into a model-level buffer."""
    cfg = _cfg("distilbert")
    model = NwayDualEncoder(cfg, share_weights=False)
    model.fuse_flat()
    enc = model.passage_encoder
    v0 = enc.flat_p._version
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    for p in model.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    assert enc.flat_p._version > v0
    v1 = enc.flat_p._version
    model.load_state_dict({k: v.clone() for k, v in model.state_dict().items()})
    assert enc.flat_p._version > v1
    w = enc.w("transformer.layer.0.ffn.lin1.weight")
    assert w.data_ptr() == dict(enc.named_parameters())["transformer.layer.0.ffn.lin1.weight"].data_ptr()


def test_zero_grad_set_to_none_clears_the_flat_gradient():
    cfg = _cfg("distilbert")
    enc = HipEncoder(cfg)
    enc.ensure_grads()
    enc.flat_g.fill_(3.0)
    torch.optim.SGD(enc.parameters(), lr=0.1).zero_grad()          # set_to_none=True is torch's default
    assert all(p.grad is None for p in enc.parameters())
    enc.ensure_grads(check_all=True)
    assert float(enc.flat_g.abs().max()) == 0.0
    assert all(p.grad is not None and p.grad.data_ptr() == enc.g(n).data_ptr() for n, p in enc.named_flat())
    enc.flat_g.fill_(2.0)
    enc.ensure_grads(check_all=True)                                  # views in place: nothing is zeroed (second tape of a step)
    assert float(enc.flat_g.min()) == 2.0


def test_token_cache_metadata_and_atomic_files(tmp_path):
    from cldrd_amd.dataset.nway_dataset import TokenCache

    class Tok:
        name_or_path = "toy"

        def __call__(self, texts, **kw):
            return {"input_ids": [[1] + [3 + (ord(c) % 7) for c in t][: kw["max_length"] - 2] + [2] for t in texts]}

    table = {7: "alpha beta", 9: "gamma", 11: "delta epsilon zeta"}
    c = TokenCache.build(table, Tok(), 8)
    stem = str(tmp_path / "queries")
    c.save(stem, {"max_len": 8, "tokenizer": "toy"})
    assert TokenCache.exists(stem) and not [f for f in os.listdir(tmp_path) if ".tmp" in f]
    d = TokenCache.load(stem, expect={"max_len": 8, "tokenizer": "toy"})
    assert np.array_equal(d.ids, c.ids) and np.array_equal(d.keys, c.keys)
    with pytest.raises(ValueError):
        TokenCache.load(stem, expect={"max_len": 16, "tokenizer": "toy"})
    with pytest.raises(ValueError):
        TokenCache.load(stem, expect={"max_len": 8, "tokenizer": "other"})


def test_common_steps_single_process():
    assert T.common_steps_per_epoch(17, False) == 17
