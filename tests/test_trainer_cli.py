"""Command line of cldrd_amd.trainer.nway_listwise (reference trainer/multistep-curriculum/nway_listwise_1.py): flag names and
defaults, the log file format with the reference's first-call quirk, and (GPU) a short run with checkpoint + resume."""
import os

import numpy as np
import pytest
import torch

from cldrd_amd.trainer import nway_listwise as T


def test_reference_flags_and_defaults():
    a = T.get_args([])
    ref = dict(learning_rate=7e-6, weight_decay=0.01, adam_epsilon=1e-8, max_grad_norm=1.0, num_train_epochs=4, warmup_steps=4000,
               reg_lambda=0.0, query_max_len=30, passage_max_len=256, train_batch_size=8, label_mode="8", logging_steps=50,
               evaluate_steps=10000, share_weights=False, in_batch_loss=False, all_in_batch_neg=False, n_gpu=1, local_rank=-1,
               model_name_or_path="sebastian-hofstaetter/distilbert-dot-tas_b-b256-msmarco",
               tokenizer_name_or_path="distilbert-base-uncased", resume=None, model_checkpoint=None, loss="lambda_mrr")
    for k, v in ref.items():
        assert getattr(a, k) == v, k
    # run_folder / log_dir / model_save_dir are joined under experiment_folder (reference get_args :143-147)
    b = T.get_args(["--experiment_folder", "/tmp/x", "--run_folder", "r1"])
    assert b.run_folder == "/tmp/x/r1" and b.log_dir == "/tmp/x/r1/log/" and b.model_save_dir == "/tmp/x/r1/models"


def test_train_log_format_and_first_call_quirk(tmp_path):
    f = str(tmp_path / "train_logs.log")
    T.write_train_logs(1, 50, 1.23456, 0.5, 0.75, 7e-6, filename=f, cutoff=10)          # header only (reference :78-84)
    T.write_train_logs(1, 100, 1.23456, 0.5, 0.75, 6.9e-6, filename=f, cutoff=10)
    lines = open(f).read().splitlines()
    assert lines[0] == "epoch\tstep\tloss_val\tmrr@10\trecall@10\tlr"
    assert lines[1] == "1\t100\t1.235\t0.500\t0.750\t0.0000069000" and len(lines) == 2
    g = str(tmp_path / "reg.log")
    T.write_train_logs(1, 50, 1.0, 0.0, 0.0, 1e-6, filename=g, reg_loss=0.1, total_aux_ratio=0.2)
    T.write_train_logs(1, 100, 1.0, 0.0, 0.0, 1e-6, filename=g, reg_loss=0.1, total_aux_ratio=0.2)
    assert open(g).read().splitlines()[0].endswith("lr\treg_loss\ttotal_aux_ratio")
    assert open(g).read().splitlines()[1].endswith("\t0.100\t0.200")


def test_schedule_and_sharding_helpers():
    assert T.linear_schedule_factor(0, 4000, 100000) == 0.0 and T.linear_schedule_factor(4000, 4000, 100000) == 1.0
    assert [i for i in range(10) if T.owns_example(i, 1, 3)] == [1, 4, 7]
    assert T.no_decay("passage_encoder.embeddings.LayerNorm.weight") and T.no_decay("x.bias")
    assert not T.no_decay("query_encoder.embeddings.word_embeddings.weight")
    # substring rule of the reference (:259): DistilBERT's sa_layer_norm / output_layer_norm weights DO get weight decay
    assert not T.no_decay("passage_encoder.transformer.layer.0.sa_layer_norm.weight")


def test_lengths_come_from_the_host_mask_without_torch_reductions():
    """attach_lengths (collate functions in the loader's workers, batch_to_device for anything else): token counts of right-padded masks, none
    for a mask that is not right-padded, untouched when the loader already gave them; numpy on the tensor's memory (on a 256-CPU host torch's
    intra-op pool took 18 ms per step for this sum: profiles/r06_microbench.txt section 8)."""
    from cldrd_amd import synthetic as syn
    from cldrd_amd.dataset.nway_dataset import attach_lengths
    b = syn.nway_batch(11, 3, 5, 8, 128, vocab=512, ragged=True)
    out = attach_lengths(b)
    m = b["nway_passages"]["attention_mask"]
    assert 0 < int(m.sum()) < m.numel()
    assert out["nway_passages"]["lengths"].dtype == torch.int64 and out["nway_passages"]["lengths"].tolist() == m.sum(-1).reshape(-1).tolist()
    assert "lengths" not in b["nway_passages"]                         # the caller's dict is not modified
    assert attach_lengths(out) is out
    left = {k: (dict(v) if isinstance(v, dict) else v) for k, v in b.items()}
    left["nway_passages"]["attention_mask"] = m.flip(-1).contiguous()
    assert "lengths" not in attach_lengths(left)["nway_passages"]
    full = attach_lengths(syn.nway_batch(11, 3, 5, 8, 24, vocab=512, ragged=False))
    assert full["nway_passages"]["lengths"].tolist() == [24] * 15
    moved = T.batch_to_device(b, torch.device("cpu"))
    assert moved["nway_passages"]["lengths"].tolist() == out["nway_passages"]["lengths"].tolist()


def test_synthetic_loader_hands_out_whole_batches_from_workers():
    """--synthetic_steps: item i of the dataset IS batch i (a function of seed, rank and i), so worker processes can produce them ahead of the GPU
    like the real loader; --synthetic_fixed gives the one-shape batches of the headline benchmark; the command line caps torch's host pool."""
    from cldrd_amd import synthetic as syn
    a = T.get_args(["--synthetic_steps", "5", "--synthetic_nway", "4", "--query_max_len", "8", "--passage_max_len", "24", "--train_batch_size", "2",
                    "--loss", "kl_div", "--loader_workers", "2"])
    assert T.get_args([]).loader_workers == 4 and not a.synthetic_fixed
    a.rank, a.nranks, a.synthetic_vocab = 0, 1, 512
    ds = T._SyntheticBatches(a)
    want = syn.nway_batch(a.seed + 3, 2, 4, 8, 24, vocab=512, ragged=True, label_kind="teacher")
    got = ds[3]
    assert len(ds) == 5 and torch.equal(got["nway_passages"]["input_ids"], want["nway_passages"]["input_ids"])
    assert torch.equal(got["labels"], want["labels"]) and got["nway_passages"]["lengths"].numel() == 8
    loader = torch.utils.data.DataLoader(ds, batch_size=None, shuffle=False, num_workers=2)
    seen = [b["nway_passages"]["input_ids"] for b in loader]
    assert len(seen) == 5 and torch.equal(seen[3], want["nway_passages"]["input_ids"])
    a.synthetic_fixed = True
    assert T._SyntheticBatches(a)[0]["nway_passages"]["lengths"].tolist() == [24] * 8
    n0 = torch.get_num_threads()
    try:
        torch.set_num_threads(max(n0, 2))
        T.cap_host_threads(1)
        assert torch.get_num_threads() == 1
        T.cap_host_threads(64)                                          # a cap never raises the count
        assert torch.get_num_threads() == 1
    finally:
        torch.set_num_threads(n0)


@pytest.mark.gpu
def test_logit_norm_regulariser_matches_torch():
    from cldrd_amd import hip_ops as ops
    x = torch.randn(8, 30, device="cuda")
    y = torch.zeros(8, 30, device="cuda")
    y[:, 0] = 1.0
    loss_out, grad = ops.loss_fwd_bwd("lambda_mrr", x, y)
    base_loss, base_grad = loss_out.clone(), grad.clone()
    reg = torch.empty(1, device="cuda")
    ops.logit_norm_reg(x, 0.05, loss_out, grad, reg)
    xr = x.clone().requires_grad_(True)
    r = xr.norm(2) * 0.05
    r.backward()
    assert torch.allclose(reg, r.detach().reshape(1), rtol=1e-5)
    assert torch.allclose(loss_out[0], base_loss[0] + r.detach(), rtol=1e-5)
    assert torch.allclose(grad, base_grad + xr.grad, rtol=1e-5, atol=1e-7)


def _dataset_args(tmp_path, extra=()):
    """Command-line arguments of a run on the committed dataset fixture (12 queries, 90 passages, 10 relT + 20 neg per example) with the
    toy tokenizer saved as a HuggingFace tokenizer directory."""
    here = os.path.dirname(os.path.abspath(__file__))
    import sys
    sys.path.insert(0, here)
    from toy_tokenizer import make_tokenizer
    fix = os.path.join(here, "golden", "nway_dataset_fixture")
    tokdir = os.path.join(str(tmp_path), "tok")
    make_tokenizer().save_pretrained(tokdir)
    return ["--experiment_folder", str(tmp_path), "--run_folder", "run", "--queries_path", os.path.join(fix, "queries.tsv"),
            "--collection_path", os.path.join(fix, "collection.tsv"), "--training_path", os.path.join(fix, "train_10relT_20neg.jsonl"),
            "--label_mode", "9", "--tokenizer_name_or_path", tokdir, "--query_max_len", "6", "--passage_max_len", "16",
            "--train_batch_size", "4", "--logging_steps", "1", "--evaluate_steps", "1000", "--warmup_steps", "1", "--num_train_epochs", "2",
            "--loss", "lambda_mrr", "--loader_workers", "2"] + list(extra)


def test_dataset_loader_runs_its_collate_in_worker_processes(tmp_path):
    """build_dataloader (reference :173-245 + the DataLoader of :247): files -> NwayDataset -> tokenizer collate in `--loader_workers` processes ->
    batches with the passages' token counts attached in the worker (attach_lengths), per-rank batch size, drop_last; the same batches with and
    without the token cache."""
    a = T.get_args(_dataset_args(tmp_path))
    a.rank, a.nranks, a.distributed = 0, 1, False
    ds, loader = T.build_dataloader(a)
    assert len(ds) == 12 and len(loader) == 3 and loader.num_workers == 2
    batches = list(loader)
    assert len(batches) == 3
    seen = []
    for b in batches:
        nw = b["nway_passages"]
        assert tuple(nw["input_ids"].shape[:2]) == (4, 30) and tuple(b["labels"].shape) == (4, 30) and b["query"]["input_ids"].shape[0] == 4
        assert nw["lengths"].tolist() == nw["attention_mask"].sum(-1).reshape(-1).tolist()
        assert nw["input_ids"].shape[-1] == int(nw["lengths"].max()) <= 16                      # padded to the longest of the batch
        seen += np.asarray(b["qid"]).tolist()
    assert len(set(seen)) == 12                                                                   # every example once per epoch
    a2 = T.get_args(_dataset_args(tmp_path, ["--token_cache_dir", os.path.join(str(tmp_path), "cache")]))
    a2.rank, a2.nranks, a2.distributed = 0, 1, False
    ds2, _ = T.build_dataloader(a2)
    for i in (0, 5, 11):
        x, y = ds.collate_fn([ds[i]]), ds2.collate_fn([ds2[i]])
        for side in ("query", "nway_passages"):
            for k in ("input_ids", "attention_mask"):
                assert torch.equal(torch.as_tensor(x[side][k]), torch.as_tensor(y[side][k]))
        assert torch.equal(x["nway_passages"]["lengths"], y["nway_passages"]["lengths"])


@pytest.mark.gpu
def test_cli_on_dataset_files_trains_through_the_worker_loader(tmp_path, monkeypatch):
    """The training command line on REAL files (not --synthetic_steps): tokenizer collate in worker processes, pinned batches, packed passages
    (the fixture's passages are 5 .. 16 tokens long), two epochs, a log line per step, finite losses, every step applied."""
    from cldrd_amd.encoder import EncoderConfig
    from cldrd_amd.models import NwayDualEncoder
    cfg = EncoderConfig(arch="distilbert", vocab_size=64, dim=128, n_heads=2, hidden_dim=256, n_layers=2, max_position_embeddings=32,
                        dropout=0.1, attention_dropout=0.1)
    real = T.NwayDualEncoder
    monkeypatch.setattr(T, "NwayDualEncoder", lambda name, **kw: real(cfg if name == "tiny-test-model" else name, **kw))
    args = T.set_env(T.get_args(_dataset_args(tmp_path, ["--model_name_or_path", "tiny-test-model", "--learning_rate", "1e-3"])))
    steps = []
    orig = T.NwayTrainer.train_step

    def spy(self, batch):
        steps.append(("lengths" in batch["nway_passages"], tuple(batch["nway_passages"]["input_ids"].shape), batch["nway_passages"]["input_ids"].is_cuda))
        return orig(self, batch)
    monkeypatch.setattr(T.NwayTrainer, "train_step", spy)
    tr = T.train(args)
    assert tr.global_step == 6 and len(steps) == 6 and all(s[0] and s[2] and s[1][:2] == (4, 30) for s in steps)
    assert torch.isfinite(tr.flat_p).all().item() and tr.skipped_steps() == 0
    log = open(os.path.join(str(tmp_path), "run", "log", "train_logs.log")).read().splitlines()
    assert len(log) == 6 and [l.split("\t")[1] for l in log[1:]] == ["2", "3", "4", "5", "6"]      # the first call only writes the header
    losses = [float(l.split("\t")[2]) for l in log[1:]]
    assert all(np.isfinite(losses)) and all(l > 0 for l in losses)


@pytest.mark.gpu
def test_cli_synthetic_run_checkpoint_and_resume(tmp_path):
    common = ["--experiment_folder", str(tmp_path), "--run_folder", "run", "--synthetic_steps", "6", "--synthetic_model", "tiny",
              "--synthetic_nway", "30", "--label_mode", "9", "--passage_max_len", "32", "--query_max_len", "8", "--train_batch_size", "4",
              "--logging_steps", "2", "--evaluate_steps", "4", "--warmup_steps", "2", "--learning_rate", "1e-3", "--reg_lambda", "0.01"]
    T.main(common + ["--num_train_epochs", "1"])
    log = open(os.path.join(str(tmp_path), "run", "log", "train_logs.log")).read().splitlines()
    assert log[0].split("\t")[:6] == ["epoch", "step", "loss_val", "mrr@10", "recall@10", "lr"] and "reg_loss" in log[0]
    assert [l.split("\t")[1] for l in log[1:]] == ["4", "6"]              # step 2 only wrote the header (reference quirk)
    ck = os.path.join(str(tmp_path), "run", "models", "checkpoint_4.pth.tar")
    c = torch.load(ck, map_location="cpu", weights_only=False)
    assert c["epoch"] == 1 and c["global_step"] == 4 and all(k.startswith("module.") for k in c["state_dict"])
    assert {"optimizer", "scheduler"} <= set(c)
    t2 = T.train(T.set_env(T.get_args(common + ["--num_train_epochs", "2", "--resume", ck])))
    assert t2.global_step == 4 + 2 * 6          # start_epoch = epoch - 1 = 0 (reference :304): both epochs run again from step 4
