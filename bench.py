#!/usr/bin/env python3
"""bench.py - headline benchmark of the CL-DRD hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one full training step of BASELINE.json configs[1] on one batch of synthetic MSMARCO-shaped input
already resident in HBM: DistilBERT-6L dual encoder (two unshared towers, random init), N=32 passages per query,
per-GPU batch B=8, seq_len 128 (queries 30), kl_div loss, 16-bit MFMA compute (fp16 operands by default, the reference's own mixed-precision
mode; CLDRD_AMP=bf16 for bf16 operands, timed as an extra leg) with fp32 accumulate and master weights, dropout 0.1
active, forward + loss + backward + RCCL gradient all-reduce (N>1) + clip_grad_norm + AdamW.  Rank 0 prints ONE JSON
line: whole-job samples/s, the roofline of the dominant kernel (the NT MFMA GEMM, timed live with HIP events on the
launch stream), index-encode passages/s, and the CPU oracle timed on the host cores (N=1 only; a reported baseline,
not the target).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
_W = int(os.environ.get("WORLD_SIZE", "1") or 1)
if (_W > 1 or os.environ.get("CLDRD_FORCE_DDP", "0") == "1") and \
        (os.environ.get("CLDRD_GRAPH", "1") == "0" or os.environ.get("CLDRD_DDP_GRAPH", "1" if _W == 1 else "0") != "1"):
    # before HIP initialises, EAGER ranks of a multi-process job only: see cl-drd_amd/__init__.py (two towers on two streams next to RCCL's;
    # ranks that replay the step as a HIP graph - one-rank groups by default, multi-rank jobs with CLDRD_DDP_GRAPH=1 - are faster with four)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

PEAK_BF16_TFLOPS = 2500.0        # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md: ~2.5 PF dense)
D, DFF, NL = 768, 3072, 6


def flops_seq_fwd(L):
    """Forward FLOPs of one DistilBERT sequence, SURVEY.md section 8d: nl*(8 L d^2 + 4 L d dff + 4 L^2 d)."""
    return NL * (8 * L * D * D + 4 * L * D * DFF + 4 * L * L * D)


def flops_seq_fwd_executed(L):
    """What the build actually runs per sequence: the LAST layer only needs token 0 (CLS pooling, reference models/nway_dual_encoder.py:52,
    56,64), so it projects K and V for every token but Q, attention, out-projection, FFN for one row."""
    full = (NL - 1) * (8 * L * D * D + 4 * L * D * DFF + 4 * L * L * D)
    last = 4 * L * D * D + 2 * D * D + 4 * L * D + 2 * D * D + 4 * D * DFF
    return full + last


EVENT_STRIDE = 5


def _being_profiled():
    return any(k.startswith(("ROCPROF", "ROCP_")) or k == "HSA_TOOLS_LIB" for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")


def pmc_traffic(kernel_substr, timeout_s=150):
    """Average HBM-side bytes per launch of a kernel, measured NOW on this box: two child runs of this same script (2 training steps,
    nothing else) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, as MI355X_MICROARCH.md prescribes;
    FETCH_SIZE is doubled: gfx950 tallies the 128-B requests of wide streaming reads at 64 B; both counters are in KiB).  A profiler
    cannot attach to the timed process, hence the children; returns (bytes_per_launch or None, note)."""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    # this process itself runs under a profiler (rocprofv3 preloads its tool library): no nested profiler, the outer one has the counters
    if _being_profiled():
        return None, "skipped: this process is being profiled (nested rocprofv3 not attempted)"
    here = os.path.abspath(__file__)
    tot = {}
    tmp = tempfile.mkdtemp(prefix="cldrd_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            cmd = [exe, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "r", "--", sys.executable, here,
                   "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-index", "--no-retrieve", "--no-kernel-events", "--no-pmc", "--no-ragged", "--no-ddp1", "--no-bf16-leg"]
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
                env.pop(k, None)
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout_s)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {ctr} failed (rc {r.returncode}): {r.stderr.decode(errors='replace')[-200:]}"
            val, ids = 0.0, set()
            for f in files:
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row["Counter_Name"] == ctr and kernel_substr in row["Kernel_Name"]:
                            val += float(row["Counter_Value"])
                            ids.add(row["Dispatch_Id"])
            if not ids:
                return None, f"no {kernel_substr} dispatch in the {ctr} pass"
            tot[ctr] = (val * 1024.0 / len(ids), len(ids))
        rd, wr = 2.0 * tot["FETCH_SIZE"][0], tot["WRITE_SIZE"][0]
        return round(rd + wr), (f"rocprofv3 --pmc, child runs of this script on this box: read {rd / 1e6:.1f} MB (2 x FETCH_SIZE) + written {wr / 1e6:.1f} MB "
                                f"per launch, averaged over {tot['FETCH_SIZE'][1]} launches")
    except Exception as e:            # a profiler problem must not take the benchmark line down
        return None, f"pmc pass failed: {type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)       # >= 50 steady-state steps after >= 10 warm-up steps
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch (queries)")
    ap.add_argument("--nway", type=int, default=32)
    ap.add_argument("--seq-len", type=int, default=128)
    ap.add_argument("--q-len", type=int, default=30)
    ap.add_argument("--loss", default="kl_div")
    ap.add_argument("--dropout", type=float, default=0.1, help="experiments only: the judged workload is the HF default 0.1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-index", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc child runs that measure roofline.traffic")
    ap.add_argument("--no-retrieve", action="store_true")
    ap.add_argument("--no-ragged", action="store_true", help="skip the MSMARCO-shaped (padded / packed) extra legs")
    ap.add_argument("--no-bf16-leg", action="store_true", help="skip the extra timing of the step in the all-bf16 mode (CLDRD_AMP=bf16)")
    ap.add_argument("--no-ddp1", action="store_true", help="skip the child run of the data-parallel code path over RCCL with one rank")
    ap.add_argument("--ddp1-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--retrieve-rows", type=int, default=1105228, help="index rows per GPU (8 841 823 / 8)")
    ap.add_argument("--retrieve-queries", type=int, default=6980, help="queries searched against the shard (MS MARCO dev: 6980)")
    args = ap.parse_args()

    if args.ddp1_child:
        return ddp1_child(args)
    import torch                      # counting devices does not initialise the GPU (nothing here may before the ranks are spawned)
    if args.gpus > torch.cuda.device_count():
        # fail fast, on every rank and before any rendezvous: a rank that cannot get its GPU must not leave the others waiting in
        # init_process_group
        raise SystemExit(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible GPUs on this node, found {torch.cuda.device_count()} "
                         f"(one process per GPU over RCCL; no scaling number can be measured here)")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N rank processes ourselves (one per GPU, RCCL), as a CHILD
        # of this process and before anything here has touched the GPU (a process that has initialised HIP must never exec).
        sys.exit(spawn_ranks(args.gpus))

    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but {world} rank(s) were launched (WORLD_SIZE={world}); refusing to report "
                         f"a number for a different GPU count")
    if torch.cuda.device_count() < (local_rank + 1):
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{local_rank} but only {torch.cuda.device_count()} device(s) are visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # the host side of every leg is kernel launches and a few tiny CPU tensor ops; torch's intra-op pool is sized to the machine (128 threads on
    # the 256-CPU GPU box: milliseconds to wake per op, profiles/r06_microbench.txt section 8, and N ranks would each own one).  cpu_baseline
    # sets its own count.
    from cldrd_amd.retriever.retrieval_utils import cap_host_threads
    cap_host_threads(8)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import cldrd_amd.synthetic as syn
    from cldrd_amd import hip_ops as ops
    from cldrd_amd.encoder import EncoderConfig
    from cldrd_amd.models import NwayDualEncoder
    from cldrd_amd.trainer import NwayTrainer

    B, N, L, Lq = args.batch, args.nway, args.seq_len, args.q_len
    cfg = EncoderConfig(arch="distilbert", dropout=args.dropout, attention_dropout=args.dropout)   # DistilBERT-6L, HF defaults
    torch.manual_seed(0)
    model = NwayDualEncoder(cfg, share_weights=False).to(dev)
    model.train()
    trainer = NwayTrainer(model, loss=args.loss, T=1.0, learning_rate=7e-6, warmup_steps=4000, total_steps=100000)
    batch = syn.nway_batch(4680 + 1000 * rank, B, N, Lq, L, ragged=False, label_kind="teacher")
    batch = {k: ({kk: vv.to(dev) for kk, vv in v.items()} if isinstance(v, dict) else v.to(dev)) for k, v in batch.items()}

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the timed region: W warm-up steps, then exactly K steps between barriers.  At N = 1 the trainer replays the step as a HIP graph
    # after its first three eager steps (trainer/nway_listwise.py: train_step); data-parallel ranks run it eagerly unless CLDRD_DDP_GRAPH=1.
    for _ in range(args.warmup):
        trainer.train_step(batch)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss_out = trainer.train_step(batch)
    sync_all()
    dt = time.perf_counter() - t0
    amp16 = bool(trainer.amp16)
    graph_replay = bool(getattr(trainer, "_graphs", None)) and any(e["graph"] is not None for e in trainer._graphs.values())
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    samples_per_s = world * B * args.steps / dt
    final_loss = float(loss_out[0].item())

    # ---- live per-launch timing of the dominant kernel (HIP events on the launch stream).  A kernel inside a replayed graph cannot be
    # bracketed by events, so the brackets go around the SAME K steps run eagerly right after the timed region (same process, same
    # buffers, same clocks; CLDRD_GRAPH=0 for this pass only) - the committed rocprofv3 summary is of the graph-replay region itself.
    gemm_events = []
    if not args.no_kernel_events:
        raw_gemm = ops.gemm_nt

        def timed_gemm(A, Bm, out, M=None, **kw):
            m = A.shape[0] if M is None else M
            # the dominant kernel is gemm_nt_ring_kernel: large-M launches (dispatch rule of cldrd_gemm_nt_ring_dispatch);
            # the query tower's M = 240 launches run the small-tile kernel on the side stream and are not part of it.
            # Every EVENT_STRIDE-th such launch of the pass is bracketed by HIP events (the stride is coprime to
            # the 42 launches of a step, so every call site is sampled): an event pair per launch costs ~0.5 ms per step
            ring = m >= 1024 and (Bm.shape[0] % 192 == 0 or Bm.shape[0] % 256 == 0) and Bm.shape[1] % 64 == 0
            if not (timed_gemm.on and ring):
                return raw_gemm(A, Bm, out, M, **kw)
            timed_gemm.count += 1
            if timed_gemm.count % EVENT_STRIDE:
                return raw_gemm(A, Bm, out, M, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = raw_gemm(A, Bm, out, M, **kw)
            e1.record()
            gemm_events.append((e0, e1, 2.0 * m * Bm.shape[0] * Bm.shape[1]))
            return r
        timed_gemm.on = False
        timed_gemm.count = 0
        ops.gemm_nt = timed_gemm
        old_graph = os.environ.get("CLDRD_GRAPH")
        os.environ["CLDRD_GRAPH"] = "0"
        try:
            for _ in range(3):
                trainer.train_step(batch)
            sync_all()
            timed_gemm.on = True
            t0e = time.perf_counter()
            for _ in range(args.steps):
                trainer.train_step(batch)
            sync_all()
            dt_events_pass = time.perf_counter() - t0e
            timed_gemm.on = False
        finally:
            ops.gemm_nt = raw_gemm
            if old_graph is None:
                os.environ.pop("CLDRD_GRAPH", None)
            else:
                os.environ["CLDRD_GRAPH"] = old_graph

    # ---- the same step on an MSMARCO-SHAPED batch (true lengths ~ clip(LogNormal(4.3, 0.35), 16, L): median 74 of 128 tokens; SURVEY.md
    # section 8d), padded as the reference computes it and PACKED (csrc/pack.hip: Linear / LayerNorm / weight gradients on the real tokens
    # only).  Extra keys: the headline above stays the all-ones batch the survey defines.
    ragged = None
    try:
        if args.no_ragged:
            raise RuntimeError("skipped (--no-ragged)")
        rb = syn.nway_batch(4680 + 1000 * rank, B, N, Lq, L, ragged=True, label_kind="teacher")
        lens = rb["nway_passages"]["attention_mask"].sum(-1).reshape(-1)
        rb = {k: ({kk: vv.to(dev) for kk, vv in v.items()} if isinstance(v, dict) else v.to(dev)) for k, v in rb.items()}
        rs = {}
        for tag in ("padded", "packed"):
            if tag == "packed":
                rb["nway_passages"]["lengths"] = lens
            for _ in range(5):
                trainer.train_step(rb)
            sync_all()
            t1 = time.perf_counter()
            for _ in range(10):
                trainer.train_step(rb)
            sync_all()
            rs[tag] = time.perf_counter() - t1
        tr_ = torch.tensor([rs["padded"], rs["packed"]], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tr_, op=dist.ReduceOp.MAX)
        ragged = {"token_fill": round(float(lens.sum()) / (B * N * L), 3), "padded_samples_per_s": round(world * B * 10 / float(tr_[0]), 1),
                  "packed_samples_per_s": round(world * B * 10 / float(tr_[1]), 1), "steps": 10}
    except Exception as exc:
        ragged = None if args.no_ragged else {"error": f"{type(exc).__name__}: {exc}"[:300]}

    # ---- the same step with BF16 operands in EVERY MFMA (CLDRD_AMP=bf16, the all-bf16 mode of round 6; BASELINE.json words cfg2 as
    # "bf16").  The headline runs the framework's default, fp16 operands with a loss-scaled backward - what the reference itself trains in
    # (trainer/multistep-curriculum/nway_listwise_1.py:129,328-352: use_fp16 default True, amp.autocast + GradScaler); both formats are 16-bit
    # operands at the same MFMA rate, fp32 accumulate / residual stream / gradient stream / master weights.  One GPU only.
    bf16_mode = None
    if rank == 0 and world == 1 and not args.no_bf16_leg and not _being_profiled():     # (a profile of this command is of the default mode only)
        try:
            old_amp = os.environ.get("CLDRD_AMP")
            os.environ["CLDRD_AMP"] = "bf16"
            try:
                torch.manual_seed(0)
                model_b = NwayDualEncoder(cfg, share_weights=False).to(dev)
                model_b.train()
                trainer_b = NwayTrainer(model_b, loss=args.loss, T=1.0, learning_rate=7e-6, warmup_steps=4000, total_steps=100000)
                assert not trainer_b.amp16
                for _ in range(max(args.warmup, 5)):
                    trainer_b.train_step(batch)
                torch.cuda.synchronize()
                tb = time.perf_counter()
                for _ in range(args.steps):
                    lb_ = trainer_b.train_step(batch)
                torch.cuda.synchronize()
                db = time.perf_counter() - tb
            finally:
                if old_amp is None:
                    os.environ.pop("CLDRD_AMP", None)
                else:
                    os.environ["CLDRD_AMP"] = old_amp
            bf16_mode = {"samples_per_s": round(B * args.steps / db, 2), "ms_per_step": round(1e3 * db / args.steps, 3), "final_loss": float(lb_[0].item()),
                         "mode": "CLDRD_AMP=bf16: EVERY MFMA operand bf16 - forward, tape and backward of both towers (the wording of BASELINE.json cfg2); "
                                 "fp32 accumulate / residual stream / gradient-path arithmetic / master weights, no loss scale"}
            del trainer_b, model_b
            torch.cuda.empty_cache()
        except Exception as exc:
            bf16_mode = {"error": f"{type(exc).__name__}: {exc}"[:300]}

    roofline = None
    if gemm_events and rank == 0:
        # an event pair around NOTHING on the same stream: what the bracket itself adds to every sample (the second event's
        # timestamp is written by a packet of its own behind the kernel); median of 64 pairs, subtracted from every sample
        empty = []
        for _ in range(64):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            e1.record()
            empty.append((e0, e1))
        torch.cuda.synchronize()
        bracket_ms = sorted(e0.elapsed_time(e1) for e0, e1 in empty)[len(empty) // 2]
        raw_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in gemm_events)
        tot_ms = raw_ms - bracket_ms * len(gemm_events)
        tot_fl = sum(f for _, _, f in gemm_events)
        achieved = tot_fl / (tot_ms * 1e-3) / 1e12
        roofline = {"kernel": "gemm_nt_ring_kernel (16-bit MFMA, fp16 or bf16 operands at the same rate; forward + data-gradient Linear GEMMs of the passage tower)",
                    "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": None, "traffic_source": "not measured (--no-pmc or N > 1)",
                    "launches_timed": len(gemm_events), "launches": timed_gemm.count, "avg_launch_us": round(1e3 * tot_ms / len(gemm_events), 2),
                    "avg_bracket_us": round(1e3 * raw_ms / len(gemm_events), 2), "empty_bracket_us": round(1e3 * bracket_ms, 2),
                    "gflop_per_launch": round(tot_fl / len(gemm_events) / 1e9, 2),
                    "time_share_of_step": round(tot_ms * 1e-3 * timed_gemm.count / len(gemm_events) / dt_events_pass, 3),
                    "events_pass": (f"the same {args.steps} steps run eagerly right after the timed region, every {EVENT_STRIDE}th launch bracketed by HIP events "
                                    f"({1e3 * dt_events_pass / args.steps:.3f} ms/step with the brackets); the timed region "
                                    + ("replays a HIP graph, whose kernels cannot be bracketed" if graph_replay else "was eager as well"))}

    # ---- index path: encode passages/s (retriever/index_text.py: bs = 512) ----
    index = None
    try:
        if not args.no_index:
            model.eval()
            ib = syn.seq_batch(99 + rank, 512, L)["seq"]
            ids, mask = ib["input_ids"].to(dev), ib["attention_mask"].to(dev)
            with torch.no_grad():
                for _ in range(2):
                    model.passage_embs({"input_ids": ids, "attention_mask": mask})
                sync_all()
                t1 = time.perf_counter()
                it = 10
                for _ in range(it):
                    emb = model.passage_embs({"input_ids": ids, "attention_mask": mask})
                torch.cuda.synchronize()
                di = time.perf_counter() - t1
            index = {"seconds": di, "it": it}
            if not args.no_ragged:
                # MSMARCO-shaped lengths, padded to the longest row of the batch as the tokenizer does, packed vs computed on the padding
                rbq = syn.seq_batch(199 + rank, 512, L, ragged=True)["seq"]
                ilen = rbq["attention_mask"].sum(-1)
                longest = int(ilen.max())
                rid, rmask = rbq["input_ids"][:, :longest].contiguous().to(dev), rbq["attention_mask"][:, :longest].contiguous().to(dev)
                with torch.no_grad():
                    for tag, extra in (("padded", {}), ("packed", {"lengths": ilen})):
                        enc = {"input_ids": rid, "attention_mask": rmask, **extra}
                        for _ in range(2):
                            model.passage_embs(enc)
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                        for _ in range(it):
                            model.passage_embs(enc)
                        torch.cuda.synchronize()
                        index[f"ragged_{tag}_s"] = time.perf_counter() - t1
                index["ragged_fill"] = float(ilen.sum()) / (512 * longest)
            # the reference's own index length: retriever/index_text.py:37 defaults max_length = 256 ("L = 256 fixed (worst
            # case) and MSMARCO-shaped").  512 passages x 256 tokens, all-ones mask; and MSMARCO-shaped lengths truncated at 256, padded to the
            # longest of the batch (what the tokenizer's padding=True gives) and packed.
            L2 = 256
            ib2 = syn.seq_batch(299 + rank, 512, L2)["seq"]
            ids2, mask2 = ib2["input_ids"].to(dev), ib2["attention_mask"].to(dev)
            with torch.no_grad():
                for _ in range(2):
                    model.passage_embs({"input_ids": ids2, "attention_mask": mask2})
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                it2 = 6
                for _ in range(it2):
                    model.passage_embs({"input_ids": ids2, "attention_mask": mask2})
                torch.cuda.synchronize()
                index["l256_s"], index["l256_it"] = time.perf_counter() - t1, it2
                if not args.no_ragged:
                    rb2 = syn.seq_batch(399 + rank, 512, L2, ragged=True)["seq"]
                    il2 = rb2["attention_mask"].sum(-1)
                    lg2 = int(il2.max())
                    rid2, rm2 = rb2["input_ids"][:, :lg2].contiguous().to(dev), rb2["attention_mask"][:, :lg2].contiguous().to(dev)
                    for tag, extra in (("padded", {}), ("packed", {"lengths": il2})):
                        enc = {"input_ids": rid2, "attention_mask": rm2, **extra}
                        for _ in range(2):
                            model.passage_embs(enc)
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                        for _ in range(it2):
                            model.passage_embs(enc)
                        torch.cuda.synchronize()
                        index[f"l256_ragged_{tag}_s"] = time.perf_counter() - t1
                    index["l256_ragged_fill"], index["l256_longest"] = float(il2.sum()) / (512 * lg2), lg2
                    # the same MSMARCO-shaped passages batched the way retriever/index_text.py batches a token cache since round 6: 8 192 rows cut
                    # into LENGTH BUCKETS (dataset.CachedSequenceDataset.length_buckets: padded size <= 65 536 token slots per chunk) instead of
                    # 16 batches of 512 consecutive rows - same embeddings bit for bit, the attention kernels' padded layout nearly all real tokens
                    from cldrd_amd.dataset import CachedSequenceDataset
                    rb3 = syn.seq_rows(499 + rank, 0, 8192, L2, ragged=True)["seq"]
                    m3 = rb3["attention_mask"]
                    il3 = m3.sum(-1).numpy()
                    chunks = []
                    for rows in CachedSequenceDataset.length_buckets(il3, 8192, 65536, 2048):
                        w3 = int(il3[rows].max())
                        rt = torch.from_numpy(rows)
                        chunks.append({"input_ids": (rb3["input_ids"][rt, :w3] * m3[rt, :w3]).contiguous().to(dev), "attention_mask": m3[rt, :w3].contiguous().to(dev),
                                       "lengths": il3[rows].tolist()})
                    for c in chunks:
                        model.passage_embs(c)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(2):
                        for c in chunks:
                            model.passage_embs(c)
                    torch.cuda.synchronize()
                    index["l256_bucketed_s"], index["l256_bucketed_rows"], index["l256_bucketed_chunks"] = time.perf_counter() - t1, 2 * 8192, len(chunks)
    except Exception as exc:      # a secondary leg must not take the headline line down with it
        index = {"error": f"{type(exc).__name__}: {exc}"[:300]}
    if not args.no_index:         # the collective sits outside the try: every rank reaches it whatever happened above
        ti = torch.tensor([index.get("seconds", float("inf"))], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(ti, op=dist.ReduceOp.MAX)
        if math.isfinite(float(ti.item())) and "error" not in index:
            pps = world * 512 * index["it"] / float(ti.item())
            extra = {}
            if "ragged_packed_s" in index:
                extra = {"msmarco_shaped": {"token_fill_of_padded_batch": round(index["ragged_fill"], 3),
                                            "padded_passages_per_s": round(world * 512 * index["it"] / index["ragged_padded_s"], 1),
                                            "packed_passages_per_s": round(world * 512 * index["it"] / index["ragged_packed_s"], 1)}}
            l256 = None
            if "l256_s" in index:
                t2 = torch.tensor([index["l256_s"]], dtype=torch.float64, device=dev)
                if world > 1:
                    dist.all_reduce(t2, op=dist.ReduceOp.MAX)
                pps2 = world * 512 * index["l256_it"] / float(t2.item())
                l256 = {"value": round(pps2, 1), "unit": "passages/s", "batch": 512, "seq_len": 256,
                        "mfma_frac": round(pps2 * flops_seq_fwd(256) / (world * PEAK_BF16_TFLOPS * 1e12), 4),
                        "note": "the reference's index length (retriever/index_text.py:37 max_length = 256), every passage 256 tokens: the worst case"}
                if "l256_ragged_packed_s" in index:
                    l256["msmarco_shaped"] = {"token_fill_of_padded_batch": round(index["l256_ragged_fill"], 3), "longest": index["l256_longest"],
                                              "padded_passages_per_s": round(world * 512 * index["l256_it"] / index["l256_ragged_padded_s"], 1),
                                              "packed_passages_per_s": round(world * 512 * index["l256_it"] / index["l256_ragged_packed_s"], 1)}
                    if "l256_bucketed_s" in index:
                        l256["msmarco_shaped"]["length_bucketed_passages_per_s"] = round(world * index["l256_bucketed_rows"] / index["l256_bucketed_s"], 1)
                        l256["msmarco_shaped"]["length_buckets"] = f"8192 passages in {index['l256_bucketed_chunks']} chunks of <= 65536 padded token slots (what index_text does with a token cache)"
            index = {"value": round(pps, 1), "unit": "passages/s", "batch": 512, "seq_len": L,
                     "mfma_frac": round(pps * flops_seq_fwd(L) / (world * PEAK_BF16_TFLOPS * 1e12), 4), **extra, "l256": l256}
        elif "error" not in index:
            index = {"error": "another rank failed in the index leg"}

    # ---- retrieve path: cfg5, one shard per GPU (8 841 823 / 8 rows x 768), ALL 6980 queries in 128-query batches, k = 1000 ----
    retrieve = None
    try:
        if not args.no_retrieve:
            from cldrd_amd.retriever.retrieval_utils import FlatIPIndex
            del trainer, model
            torch.cuda.empty_cache()
            rows, nq_r, kq = args.retrieve_rows, args.retrieve_queries, 1000
            gen = torch.Generator(device=dev).manual_seed(1234 + rank)
            P = torch.randn(rows, D, device=dev, generator=gen)
            P *= ((9.0 + 3.0 * torch.rand(rows, 1, device=dev, generator=gen)) / P.norm(dim=1, keepdim=True))
            flat_index = FlatIPIndex.from_device_rows(P, id_offset=rank * rows)
            qn = torch.randn(nq_r, D, device=dev, generator=gen)
            qn *= (10.0 / qn.norm(dim=1, keepdim=True))
            flat_index.search_device(qn, kq)                   # warm-up: one full search (workspaces, and the clocks settle: the first
                                                               # ~10 passes after an idle gap run 20-30 % slower than the steady state)
            sync_all()
            nb = (nq_r + 127) // 128                          # reference batches of 128 queries (the unit of SURVEY.md section 8d)
            # (1) device-resident search: queries and results stay in HBM, one host sync (the proof flags)
            t2 = time.perf_counter()
            Dq, Iq, st = flat_index.search_device(qn, kq)
            torch.cuda.synchronize()
            dr = time.perf_counter() - t2
            # (2) the same with HIP events around the enqueued pipeline, and candidate statistics
            flat_index.profile = True
            _, _, st = flat_index.search_device(qn, kq)
            flat_index.profile = False
            # (3) the reference-shaped host API (numpy in, numpy out: PCIe both ways + id mapping), then the run file of those results
            # (retrieve_top_passages.py:90-107: 6980 x 1000 lines) through the native writer - the host tail of the CLI, for the record
            qh = qn.cpu().numpy()
            sync_all()
            t3 = time.perf_counter()
            Dh, Ih = flat_index.search(qh, kq)
            dh = time.perf_counter() - t3
            run_file_s = None
            if rank == 0:
                import tempfile
                from cldrd_amd.retriever.retrieve_top_passages import write_run_file
                with tempfile.TemporaryDirectory(dir="/tmp") as td:
                    t4 = time.perf_counter()
                    write_run_file(os.path.join(td, "dev.run"), list(range(nq_r)), Ih, Dh)
                    run_file_s = time.perf_counter() - t4
            del Dh, Ih
            # (3b) the 8-way sharded search END TO END (north_star: "per-shard brute-force top-k GEMM then a host-side merge"; the reference's
            # dead faiss branch retriever/retrieval_utils.py:164-182): every rank searches its shard, the lists go to rank 0 with one
            # dist.gather per tensor over RCCL and rank 0 merges them on the device (ShardedFlatIPIndex.gather_merge_device) - search,
            # gather AND merge inside the wall clock.  At N = 1 there is nobody to gather from: the merge alone is timed on eight
            # synthetic shard lists of the cfg5 shape (this shard's own lists, scores shifted per shard so that the lists interleave).
            from cldrd_amd.retriever.retrieval_utils import ShardedFlatIPIndex
            sharded = {}
            if world > 1:
                sh = ShardedFlatIPIndex(flat_index, rank, world)
                for rep in range(2):                          # the first pass warms the communicator up
                    sync_all()
                    t6 = time.perf_counter()
                    Dd_, Id_ = flat_index.search_ids_device(qn, kq)
                    Dm_, Im_ = sh.gather_merge_device(Dd_.contiguous(), Id_.contiguous(), kq)
                    torch.cuda.synchronize()
                    d_sh = time.perf_counter() - t6
                ts_ = torch.tensor([d_sh], dtype=torch.float64, device=dev)
                dist.all_reduce(ts_, op=dist.ReduceOp.MAX)
                d_sh = float(ts_.item())
                if rank == 0:
                    assert bool((Dm_[:, 1:] <= Dm_[:, :-1]).all()) and int(Im_.min()) >= 0 and int(Im_.max()) < world * rows
                sharded = {"sharded_wall_s": round(d_sh, 4), "sharded_wall_ms_per_batch": round(1e3 * d_sh / nb, 3), "sharded_merge": sh.last_merge.get("path"),
                           "sharded_queries_per_s": round(nq_r / d_sh, 1),
                           # sum of the bytes all shards stream (SURVEY.md 8d unit x batches x shards) / (wall x n_gpu x HBM peak)
                           "sharded_wall_hbm_frac": round((rows * D * 2 + 128 * D * 2) / (1e3 * d_sh / nb) / 1e6 / 8000.0, 4)}
                del Dd_, Id_, Dm_, Im_
            else:
                Id64 = Iq.to(torch.int64)
                allD = torch.stack([Dq + 0.37 * r_ / 8.0 for r_ in range(8)]).contiguous()
                allI = torch.stack([torch.where(Id64 >= 0, Id64 + r_ * rows, Id64) for r_ in range(8)]).contiguous()
                ops2m = ops.merge_topk_device
                ops2m(allD[:, :64].contiguous(), allI[:, :64].contiguous(), kq)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                Dm_, Im_ = ops2m(allD, allI, kq)
                e1.record()
                torch.cuda.synchronize()
                hD, hI = list(allD.cpu().numpy()), list(allI.cpu().numpy())
                ops.merge_topk_host([d_[:8] for d_ in hD], [i_[:8] for i_ in hI], kq)
                t7 = time.perf_counter()
                Dh8, Ih8 = ops.merge_topk_host(hD, hI, kq)
                d_h8 = time.perf_counter() - t7
                import numpy as _np
                assert _np.array_equal(Ih8, Im_.cpu().numpy()) and _np.array_equal(Dh8, Dm_.cpu().numpy()), "device and host merge disagree"
                sharded = {"merge8_ms": round(e0.elapsed_time(e1), 3), "merge8_host_ms": round(1e3 * d_h8, 2),
                           "merge8": f"eight synthetic shard lists of {nq_r} x {kq} (this shard's lists, scores shifted per shard) merged on the device "
                                     "(cldrd_merge_topk_device, HIP events) and by the native host merge (cldrd_merge_topk, wall clock); results identical"}
                del allD, allI, Dm_, Im_, hD, hI, Dh8, Ih8, Id64
            # (4) the dominant kernel alone: the fp16 streaming scan of one 128-query batch, HIP events on its stream
            from cldrd_amd import hip_ops as ops2
            qh16 = qn[:128].half().contiguous()
            thr = torch.full((128,), 11.5, device=dev)
            QT = flat_index.query_tile                      # queries per pass over the index: 256 at d = 768 (two reference batches)
            qh16 = qn[:QT].half().contiguous()
            thr = torch.full((QT,), 11.9, device=dev)      # ~1000 hits per query, as in the search
            counts = torch.zeros(QT + 1, dtype=torch.int32, device=dev)
            cr = torch.empty(QT, 8192, dtype=torch.int32, device=dev)
            cs_ = torch.empty(QT, 8192, dtype=torch.float32, device=dev)
            for _ in range(40):                               # same settling as above before the timed launches
                ops2.topk_scan_filter(qh16, flat_index._p16, thr, counts, cr, cs_)
            evs = []
            for _ in range(20):
                counts.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops2.topk_scan_filter(qh16, flat_index._p16, thr, counts, cr, cs_)
                e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            scan_pass_ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
            scan_ms = scan_pass_ms * 128.0 / QT             # per 128-query reference batch
            scan_bytes = rows * D * 2 + 128 * D * 2            # bytes of one 128-query batch x one shard scan (16-bit rows)
            batch_ms = st["search_ms"] / nb
            retrieve = {"seconds": dr, "nq": nq_r, "rows_per_shard": rows, "rows_total": world * rows, "k": kq, "batch": 128,
                        "scans": st["scans"], "rescans": st["rescans"], "unproven_first_pass": st["unproven_first_pass"],
                        "candidates_per_query": round(st["candidates"] / nq_r, 1), "rescored_per_query": round(st["rescored"] / nq_r, 1),
                        "queries_per_pass": QT, "scan_pass_ms": round(scan_pass_ms, 3),
                        # what the scan kernel physically does per pass of QT queries: bytes streamed once, and its MFMA work - it sits on
                        # NEITHER roof (issue-bound: LDS-DMA issue + MFMA of two waves per SIMD); `scan_hbm_frac` below is the ALGORITHMIC
                        # figure of SURVEY.md section 8d (bytes of one 128-query reference batch per shard scan / time per such batch)
                        "scan_physical_hbm_gb_s": round((rows * D * 2 + QT * D * 2) / scan_pass_ms / 1e6, 1),
                        "scan_physical_hbm_frac": round((rows * D * 2 + QT * D * 2) / scan_pass_ms / 1e6 / 8000.0, 4),
                        "scan_mfma_frac": round(2.0 * QT * rows * D / scan_pass_ms / 1e9 / PEAK_BF16_TFLOPS, 4),
                        "scan_bound": "issue (neither roof): physical HBM and fp16 MFMA fractions above; the algorithmic HBM fraction counts the index bytes once per 128-query batch",
                        "scan_kernel_ms": round(scan_ms, 3), "scan_hbm_gb_s": round(scan_bytes / scan_ms / 1e6, 1),
                        "scan_hbm_frac": round(scan_bytes / scan_ms / 1e6 / 8000.0, 4),
                        "scan_tflops": round(2.0 * 128 * rows * D / scan_ms / 1e9, 1),
                        "path_ms_per_batch": round(batch_ms, 3), "path_hbm_gb_s": round(scan_bytes / batch_ms / 1e6, 1),
                        "path_hbm_frac": round(scan_bytes / batch_ms / 1e6 / 8000.0, 4),
                        "wall_ms_per_batch": round(1e3 * dr / nb, 3), "wall_hbm_frac": round(scan_bytes / (1e3 * dr / nb) / 1e6 / 8000.0, 4),
                        "host_api_queries_per_s": round(nq_r / dh, 1), "host_api_s": round(dh, 4),
                        "run_file_s": None if run_file_s is None else round(run_file_s, 4), "run_file_lines": nq_r * kq, **sharded}
            del flat_index, P
            # (5) the same search on a CLS-LIKE (anisotropic) shard: the regime of real dual-encoder embeddings, where every row scores
            # close to every other, the 2 eps band under the k-th score holds ~3x the rows and the re-score is the larger part of a pass
            # (the corpus of tests/test_gpu_retrieval.py::test_cls_like_anisotropic_corpus_at_shard_size, all 6 980 queries)
            try:
                torch.cuda.empty_cache()
                Pc, u_c = syn.cls_like_corpus(rows, D, 777 + rank, dev)
                qc = syn.cls_like_queries(nq_r, u_c, 778 + rank)
                cidx = FlatIPIndex.from_device_rows(Pc, id_offset=rank * rows)
                cidx.search_device(qc, kq)
                sync_all()
                t5 = time.perf_counter()
                cidx.search_device(qc, kq)
                torch.cuda.synchronize()
                dc = time.perf_counter() - t5
                cidx.profile = True
                _, _, stc = cidx.search_device(qc, kq)
                cb_ms = stc["search_ms"] / nb
                retrieve["cls_like"] = {"queries_per_s": round(nq_r / dc, 1), "scans": stc["scans"], "rescans": stc["rescans"],
                                        "unproven_first_pass": stc["unproven_first_pass"], "fallback_queries": stc.get("fallback_queries", 0),
                                        "candidates_per_query": round(stc["candidates"] / nq_r, 1), "rescored_per_query": round(stc["rescored"] / nq_r, 1),
                                        "cap2": stc.get("cap2"), "path_ms_per_batch": round(cb_ms, 3),
                                        "path_hbm_frac": round(scan_bytes / cb_ms / 1e6 / 8000.0, 4),
                                        "wall_ms_per_batch": round(1e3 * dc / nb, 3), "wall_hbm_frac": round(scan_bytes / (1e3 * dc / nb) / 1e6 / 8000.0, 4),
                                        "corpus": "rows = u x 3.6 U(0.8, 1.2) + 0.045 N(0, I), queries 4.6 U(0.9, 1.1) u + 0.06 N(0, I): score std / mean 0.12"}
                del cidx, Pc
            except Exception as exc:
                retrieve["cls_like"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
    except Exception as exc:      # a secondary leg must not take the headline line down with it
        retrieve = {"error": f"{type(exc).__name__}: {exc}"[:300]}
    if not args.no_retrieve:      # every rank searches its own shard for the same queries: the slowest one sets the rate
        tr_ = torch.tensor([retrieve.get("seconds", float("inf"))], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tr_, op=dist.ReduceOp.MAX)
        if math.isfinite(float(tr_.item())) and "error" not in retrieve:
            dr = float(tr_.item())
            nq_ = retrieve.pop("nq")
            retrieve.pop("seconds")
            retrieve = {"queries_per_s": round(nq_ / dr, 1), "row_scans_per_s": round(retrieve["rows_total"] * nq_ / dr, 1), **retrieve}
        elif "error" not in retrieve:
            retrieve = {"error": "another rank failed in the retrieve leg"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(N, L, Lq)
        except Exception as exc:
            cpu = {"error": f"{type(exc).__name__}: {exc}"[:300]}

    # The data-parallel CODE PATH with one rank over the real backend (ProcessGroupNCCL = RCCL): constructor broadcast, per-bucket
    # all_reduce(async_op=True) from the backward hooks on the communication stream, Work handles, weight gradients flushed every 3 layers,
    # eager launches.  Not a scaling number (no second GPU exists on this box): the per-rank cost of that path next to the headline's.
    ddp1 = None
    if rank == 0 and world == 1 and not args.no_ddp1:
        torch.cuda.synchronize()
        ddp1 = ddp1_parent(args)

    # roofline.traffic, last (the model and every buffer of this process stay allocated, but nothing of ours runs meanwhile): two
    # rocprofv3 --pmc child runs, rank 0 at N = 1 only
    if rank == 0 and world == 1 and roofline is not None and not args.no_pmc:
        torch.cuda.synchronize()
        roofline["traffic"], roofline["traffic_source"] = pmc_traffic("gemm_nt_ring_kernel")

    if rank == 0:
        flops_per_sample = 3.0 * (N * flops_seq_fwd(L) + flops_seq_fwd(Lq))
        exec_per_sample = 3.0 * (N * flops_seq_fwd_executed(L) + flops_seq_fwd_executed(Lq))
        out = {
            "metric": "train (q,N-psg) samples/sec + index passages/sec, DistilBERT N=32 at 1/8 GPUs",
            "value": round(samples_per_s, 2), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "step_launch": "hip graph replay" if graph_replay else "eager",
            "dtype": "fp16" if amp16 else "bf16", "data": "synthetic",
            "config": {"workload": f"cfg2: DistilBERT-6L dual encoder (2 unshared towers), N={N}, {args.loss}, seq_len={L}, q_len={Lq}, "
                                   f"per-GPU batch {B}, dropout {args.dropout:g}, fwd+loss+bwd+allreduce+clip+AdamW; "
                                   + ("fp16 MFMA operands (16-bit, same MFMA rate as bf16; the reference's own use_fp16 autocast mode), loss-scaled backward, "
                                      if amp16 else "bf16 MFMA operands in every GEMM and in attention, ")
                                   + "fp32 accumulate / residual stream / gradient stream / master weights",
                       "global_batch": world * B, "parallelism": f"dp{world}"},
            "model_tflop_per_sample": round(flops_per_sample / 1e12, 4),
            "step_mfma_frac": round(samples_per_s * flops_per_sample / (world * PEAK_BF16_TFLOPS * 1e12), 4),
            # model FLOPs count the full last layer (SURVEY.md section 8d convention); the build computes only its CLS row:
            "executed_tflop_per_step": round(B * exec_per_sample / 1e12, 3),
            "step_mfma_frac_executed": round(samples_per_s * exec_per_sample / (world * PEAK_BF16_TFLOPS * 1e12), 4),
            "final_loss": final_loss,
            "bf16_operand_mode": bf16_mode,
            "msmarco_shaped_train": ragged,
            "ddp_path_one_rank_rccl": ddp1,
            "index": index, "retrieve": retrieve, "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def ddp1_parent(args, timeout_s=240):
    """Run `bench.py --ddp1-child` (its own process group) and return its JSON, or the reason it did not finish."""
    import socket
    import subprocess
    if _being_profiled():
        return {"skipped": "this process is being profiled: the child would inherit the profiler's preload environment and write into its output"}
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", CLDRD_FORCE_DDP="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.abspath(__file__), "--ddp1-child", "--batch", str(args.batch), "--nway", str(args.nway), "--seq-len",
           str(args.seq_len), "--q-len", str(args.q_len), "--loss", args.loss, "--dropout", str(args.dropout), "--steps", "20", "--warmup", "6"]
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s, text=True)
        lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": f"child rc {r.returncode}: {r.stderr.strip()[-240:]}"}
        return json.loads(lines[-1])
    except subprocess.TimeoutExpired:
        return {"error": f"child did not finish within {timeout_s} s"}
    except Exception as exc:
        return {"error": f"{type(exc).__name__}: {exc}"[:300]}


def ddp1_child(args):
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import cldrd_amd.synthetic as syn
    from cldrd_amd.encoder import EncoderConfig
    from cldrd_amd.models import NwayDualEncoder
    from cldrd_amd.trainer import NwayTrainer
    cfg = EncoderConfig(arch="distilbert", dropout=args.dropout, attention_dropout=args.dropout)
    torch.manual_seed(0)
    model = NwayDualEncoder(cfg, share_weights=False).to(dev)
    model.train()
    trainer = NwayTrainer(model, loss=args.loss, T=1.0, learning_rate=7e-6, warmup_steps=4000, total_steps=100000)
    if os.environ.get("CLDRD_FORCE_DDP", "0") == "1":
        assert trainer.distributed and trainer.comm_stream is not None
    batch = syn.nway_batch(4680, args.batch, args.nway, args.q_len, args.seq_len, ragged=False, label_kind="teacher")
    batch = {k: ({kk: vv.to(dev) for kk, vv in v.items()} if isinstance(v, dict) else v.to(dev)) for k, v in batch.items()}
    for _ in range(args.warmup):
        trainer.train_step(batch)
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss_out = trainer.train_step(batch)
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"samples_per_s": round(args.batch * args.steps / dt, 2), "ms_per_step": round(1e3 * dt / args.steps, 3), "steps": args.steps,
                      "backend": dist.get_backend(), "world_size": 1, "ddp_path": bool(trainer.distributed), "buckets": len(trainer.buckets),
                      "step_launch": "hip graph replay (RCCL collectives captured)" if any(e["graph"] is not None for e in getattr(trainer, "_graphs", {}).values()) else "eager",
                      "final_loss": float(loss_out[0].item())}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0


def spawn_ranks(n: int) -> int:
    """Run this same command line under torch.distributed.run with n ranks on this node; returns the launcher's exit code.
    Rank 0 of the children prints the JSON line; it passes through on stdout."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL across processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def cpu_baseline(N, L, Lq):
    """The CPU oracle (oracle/, a port of the reference path pinned to it by tests/golden) timed on this box's host cores: fp32
    training steps (forward + loss + backward + clip + AdamW) of the same workload (cfg2: N=32, L=128, kl_div) at B=2 - one untimed
    warm-up step, then four timed ones: a bounded sample (~10-20 s of CPU work), all cores the process may use."""
    import numpy as np
    import torch
    from oracle import encoder_ref as E
    from oracle import losses_ref as LR
    import cldrd_amd.synthetic as syn
    # threads: every core the process may use, up to 32.  Measured on the MI355X host (256 logical CPUs visible): the same two
    # steps took 496 s on 256 torch threads and 6.8 s on 32 - beyond that torch-CPU's intra-op parallelism only adds contention.
    cores = min(len(os.sched_getaffinity(0)), 32)
    torch.set_num_threads(cores)
    Bc, timed = 2, 4          # ~11 s on the MI355X host at 32 threads (the prescribed bounded sample is 10-30 s of CPU work)
    cfg = E.RefConfig()
    shapes = E.param_shapes(cfg)
    g = torch.Generator().manual_seed(0)
    qp = {k: (torch.randn(s, generator=g) * 0.02).requires_grad_(True) for k, s in shapes.items()}
    pp = {k: (torch.randn(s, generator=g) * 0.02).requires_grad_(True) for k, s in shapes.items()}
    params = list(qp.values()) + list(pp.values())
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]

    def step(batch, t):
        for p in params:
            p.grad = None
        logits = E.nway_forward(qp, pp, cfg, batch["query"], batch["nway_passages"])
        _, dl = LR.kl_div(logits.detach().numpy(), batch["labels"].numpy())
        logits.backward(torch.from_numpy(dl).float())
        with torch.no_grad():
            total = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in params)).item()
            coef = min(1.0, 1.0 / (total + 1e-6))
            for p, mm, vv in zip(params, m, v):
                gq = p.grad * coef
                mm.mul_(0.9).add_(gq, alpha=0.1)
                vv.mul_(0.999).addcmul_(gq, gq, value=0.001)
                p.addcdiv_(mm, vv.sqrt().add_(1e-8), value=-7e-6 * (1 - 0.999 ** t) ** 0.5 / (1 - 0.9 ** t))
                p.mul_(1 - 7e-6 * 0.01)

    step(syn.nway_batch(2, 1, 4, 8, 32), 1)                    # warm-up (thread pool, allocator)
    t0 = time.perf_counter()
    for i in range(timed):
        step(syn.nway_batch(10 + i, Bc, N, Lq, L), 2 + i)
    dt = time.perf_counter() - t0
    Bc = Bc * timed
    out = {"value": round(Bc / dt, 4), "unit": "samples/s", "cores": cores, "kind": "port",
           "sample": f"{timed} fp32 steps (fwd+loss+bwd+clip+AdamW) of the same workload at B={Bc // timed} (N={N}, L={L}) after 1 warm-up step, oracle/encoder_ref.py on torch-CPU, {dt:.1f} s"}
    # the other two legs of the metric on the same cores: encode 64 x L passages; exact top-1000 of 128 queries
    # over a 200 000-row fp32 shard (oracle/retrieval_ref.py: the faiss IndexFlatIP contract)
    from oracle import retrieval_ref as RR
    with torch.no_grad():
        pn = {k: t.detach() for k, t in pp.items()}
        seq = syn.seq_batch(7, 64, L)["seq"]
        E.cls_embs(pn, cfg, {k: t[:2] for k, t in seq.items()})
        t0 = time.perf_counter()
        E.cls_embs(pn, cfg, seq)
        de = time.perf_counter() - t0
    rng = np.random.default_rng(0)
    Pm = rng.standard_normal((200000, 768), dtype=np.float32)
    Qm = rng.standard_normal((128, 768), dtype=np.float32)
    t0 = time.perf_counter()
    S = Qm @ Pm.T                                              # fp32 SGEMM on the host cores, as faiss IndexFlatIP does
    part = np.argpartition(-S, 1000, axis=1)[:, :1000]
    top = np.take_along_axis(S, part, axis=1)
    order = np.lexsort((part, -top), axis=1)                   # score desc, row asc
    I_cpu = np.take_along_axis(part, order, axis=1)
    dk = time.perf_counter() - t0
    D_ref, I_ref = RR.flat_ip_search(Pm[:20000], None, Qm[:4], 50)          # the timed formulation agrees with the oracle's contract
    S4 = Qm[:4] @ Pm[:20000].T
    assert np.array_equal(np.sort(np.argpartition(-S4, 50, axis=1)[:, :50], axis=1), np.sort(I_ref, axis=1)), "cpu top-k disagrees with the oracle"
    out["encode"] = {"value": round(64 / de, 2), "unit": "passages/s", "sample": f"64 passages x {L} tokens, fp32 forward, {de:.1f} s"}
    out["topk"] = {"value": round(128 / dk, 1), "unit": "queries/s", "rows": 200000,
                   "sample": f"128 queries x 200 000 x 768 fp32 rows, top-1000 by fp32 SGEMM + argpartition + sort (numpy), {dk:.1f} s"}
    return out


if __name__ == "__main__":
    main()
