#!/usr/bin/env python3
"""cfg5 of BASELINE.json as ONE job at full size on one GPU, through the two command lines of the retrieve path:

    token cache (8 841 823 synthetic MSMARCO-shaped passages, uint16 [n, 256], memory-mapped)
      -> retriever.index_text.main   (passage tower, max_length 256, batches of 512; index file + meta.pkl)
      -> retriever.retrieve_top_passages.main   (6 980 synthetic queries, k = 1000; run file)

and the wall clock of every phase: open the cache, encode (with the host-side split: loader wait / H2D + enqueue / waiting for the GPU's
previous batch / gather), index build, index write, meta.pkl, model load, query encode, index read, attach (H2D + scan shadow), search,
run file.  Reference: retriever/index_text.py:84-109, retriever/retrieve_top_passages.py:80-107; its README quotes "~2.5 h" for the index
step on its hardware (README.md:20) - context, not a comparison.

What is synthetic: the token ids (no tokenizer vocabulary or MS MARCO offline) with MS-MARCO-like lengths (cldrd_amd.synthetic.seq_rows:
median ~74 tokens), written straight into the SequenceTokenCache file format (the one-off tokenisation is not part of the timed path); the
model is a random-init DistilBERT pair saved as a reference-style checkpoint.

    python tools/time_cfg5_full.py [--rows 8841823] [--queries 6980] [--workdir DIR] [--out profiles/r06_cfg5_full.txt]

Needs ~32 GB of disk in --workdir and ~70 GB of host memory at full size; refuses when the box has less (--force overrides).
"""
import argparse
import json
import os
import shutil
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def _gen_chunk(job):
    """worker: rows [lo, hi) of the synthetic collection into the memory-mapped cache arrays"""
    stem, lo, hi, L, seed = job
    import cldrd_amd.synthetic as syn
    ids = np.load(stem + ".ids.npy", mmap_mode="r+")
    lens = np.load(stem + ".lens.npy", mmap_mode="r+")
    step = 8192
    for a in range(lo, hi, step):
        z = min(hi, a + step)
        b = syn.seq_rows(seed, a, z - a, L, ragged=True)
        m = b["seq"]["attention_mask"].numpy()
        ids[a:z] = (b["seq"]["input_ids"].numpy() * m).astype(ids.dtype)
        lens[a:z] = m.sum(1).astype(np.int32)
    ids.flush()
    lens.flush()
    return hi - lo


def build_token_cache(stem, rows, L, seed=99, procs=8):
    import multiprocessing as mp
    ids = np.lib.format.open_memmap(stem + ".ids.npy", mode="w+", dtype=np.uint16, shape=(rows, L))
    del ids
    lens = np.lib.format.open_memmap(stem + ".lens.npy", mode="w+", dtype=np.int32, shape=(rows,))
    del lens
    np.save(stem + ".keys.npy", np.arange(rows, dtype=np.int64))
    per = -(-rows // (procs * 4))
    jobs = [(stem, lo, min(rows, lo + per), L, seed) for lo in range(0, rows, per)]
    with mp.get_context("spawn").Pool(procs) as pool:
        done = sum(pool.map(_gen_chunk, jobs))
    assert done == rows
    with open(stem + ".meta.json", "w") as fh:
        json.dump({"source": "synthetic (cldrd_amd.synthetic.seq_rows)", "max_length": int(L), "rows": int(rows), "tokenizer": "none", "vocab_size": 30522}, fh)


def mem_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return float("nan")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=8841823)
    ap.add_argument("--queries", type=int, default=6980)
    ap.add_argument("--max_length", type=int, default=256)
    ap.add_argument("--top_k", type=int, default=1000)
    ap.add_argument("--workdir", default="/tmp/cldrd_cfg5_full")
    ap.add_argument("--out", default="")
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--init_std", type=float, default=0.06)
    ap.add_argument("--bucket_window", type=int, default=16384, help="0: batches of 512 consecutive rows, as the reference forms them")
    ap.add_argument("--force", action="store_true")
    a = ap.parse_args()
    os.makedirs(a.workdir, exist_ok=True)
    need_disk = a.rows * (768 * 4 + a.max_length * 2 + 16) / 1e9 + 1.0
    need_mem = a.rows * 768 * 4 * 2.3 / 1e9 + 8.0
    free_disk = shutil.disk_usage(a.workdir).free / 1e9
    lines = [f"cfg5 end to end, one GPU: {a.rows} passages x max_length {a.max_length}, {a.queries} queries, top-{a.top_k}; batches: "
             + (f"length buckets inside windows of {a.bucket_window} rows" if a.bucket_window else "512 consecutive rows"),
             f"host: {os.cpu_count()} CPUs ({len(os.sched_getaffinity(0))} usable), {mem_gb():.0f} GB available memory, {free_disk:.0f} GB free in {a.workdir} "
             f"(needs ~{need_disk:.0f} GB disk, ~{need_mem:.0f} GB memory)"]
    print(lines[-1], flush=True)
    if not a.force and (free_disk < need_disk or mem_gb() < need_mem):
        print("not enough disk / memory for this size: pass fewer --rows (or --force)")
        return 2
    import torch
    from cldrd_amd.encoder import EncoderConfig
    from cldrd_amd.models import NwayDualEncoder
    from cldrd_amd.retriever import index_text as IT
    from cldrd_amd.retriever import retrieve_top_passages as RTP

    t_job = time.perf_counter()
    stem = os.path.join(a.workdir, f"synthetic.L{a.max_length}.seqcache")
    t0 = time.perf_counter()
    build_token_cache(stem, a.rows, a.max_length, procs=max(2, min(32, len(os.sched_getaffinity(0)) - 1)))
    t_cache = time.perf_counter() - t0
    lens = np.load(stem + ".lens.npy", mmap_mode="r")
    n_tok = int(np.asarray(lens, dtype=np.int64).sum())
    lines.append(f"token cache (synthetic, NOT a phase of the path: stands for the one-off tokenisation): {t_cache:.1f} s, {n_tok / 1e6:.0f} M tokens, "
                 f"mean length {n_tok / a.rows:.1f}, fill {n_tok / (a.rows * a.max_length):.2f} of max_length")
    print(lines[-1], flush=True)

    torch.manual_seed(0)
    # weights drawn 3 x wider than the HF init (0.06 instead of 0.02): the random encoder's CLS vectors are then as diverse as a trained dual
    # encoder's (pairwise cosine 0.87, q.p spread ~19 % of its mean; the reference model: 17 +- 2).  At the HF init scale they are
    # near-duplicates (cosine 0.993, q.p = 759 +- 0.6: a spread of 8e-4, below what ANY 16-bit scan resolves) and every query takes the exact
    # fp32 fallback: measured 110 s for the search phase (profiles/r06_cfg5_full.txt, first run)
    model = NwayDualEncoder(EncoderConfig(arch="distilbert", initializer_range=a.init_std), share_weights=False)
    mdir = os.path.join(a.workdir, "model")
    model.query_encoder.save_pretrained(mdir)
    ckpt = os.path.join(a.workdir, "checkpoint_1.pth.tar")
    torch.save({"state_dict": {"module." + k: v for k, v in model.state_dict().items()}}, ckpt)
    del model

    idir = os.path.join(a.workdir, "index")
    from cldrd_amd.retriever import retrieval_utils as RU
    prog = {"t": time.perf_counter(), "tm": {}}

    def progress(tm):          # every 500 batches (stdout only): does the loop's rate hold over 17 000 batches?
        now = time.perf_counter()
        d = {k: tm[k] - prog["tm"].get(k, 0.0) for k in ("load_s", "h2d_enqueue_s", "d2h_wait_s", "gather_s")}
        print(f"  batch {tm['batches']:6d}: {500 / (now - prog['t']):6.1f} batches/s | per batch ms: loader {2 * d['load_s']:.2f} H2D+enqueue {2 * d['h2d_enqueue_s']:.2f} "
              f"GPU wait {2 * d['d2h_wait_s']:.2f} gather {2 * d['gather_s']:.2f}", flush=True)
        prog["t"], prog["tm"] = now, dict(tm)
    RU.PROGRESS_HOOK = progress
    t0 = time.perf_counter()
    ipath = IT.main(IT.get_args(["--resume", ckpt, "--model_name_or_path", mdir, "--max_length", str(a.max_length), "--index_dir", idir,
                                 "--token_cache_stem", stem, "--loader_workers", str(a.workers), "--bucket_window", str(a.bucket_window)]))
    t_index = time.perf_counter() - t0
    RU.PROGRESS_HOOK = None
    ti = IT.main.last_timings
    host = ti.get("encode_host_phases", {})
    lines.append("")
    lines.append(f"index_text.main: {t_index:.1f} s   ({a.rows / max(ti['encode_s'], 1e-9):.0f} passages/s in the encode phase)")
    for k, label in (("model_load_s", "model load (checkpoint -> HBM, shadows)"), ("open_collection_s", "open the token cache"),
                     ("encode_s", "encode: loader -> H2D -> passage tower -> D2H -> host array"),
                     ("index_build_s", "index build (id map, FlatIPIndex over the host array)"), ("index_write_s", "index write (np.save of the fp32 rows + meta)"),
                     ("meta_pkl_s", "meta.pkl (text_ids + the reference's 8.8 M-entry dict)")):
        lines.append(f"  {label:<72s} {ti.get(k, float('nan')):9.2f} s")
    if host:
        lines.append(f"  host side of the encode loop ({host.get('batches', 0)} batches; bucket_window {a.bucket_window}): waiting for the loader {host['load_s']:.1f} s | batch H2D + enqueue "
                     f"{host['h2d_enqueue_s']:.1f} s | waiting for the GPU (encode + D2H of the previous batch) {host['d2h_wait_s']:.1f} s | gather into [n, 768] {host['gather_s']:.1f} s")
    isize = sum(os.path.getsize(os.path.join(idir, f)) for f in os.listdir(idir)) / 1e9
    lines.append(f"  index directory: {isize:.1f} GB")
    for ln in lines[-9:]:
        print(ln, flush=True)
    import gc
    gc.collect()
    torch.cuda.empty_cache()

    out = os.path.join(a.workdir, "runs", "dev.run")
    t0 = time.perf_counter()
    RTP.main(RTP.get_args(["--resume", ckpt, "--model_name_or_path", mdir, "--index_path", ipath, "--max_length", "30", "--top_k", str(a.top_k),
                           "--synthetic_queries", str(a.queries), "--output_path", out]))
    t_ret = time.perf_counter() - t0
    tr = RTP.main.last_timings
    lines.append("")
    lines.append(f"retrieve_top_passages.main: {t_ret:.1f} s")
    for k, label in (("model_load_s", "model load"), ("encode_queries_s", f"encode {a.queries} queries"), ("index_read_s", "index read (memory map)"),
                     ("index_to_gpu_s", "attach: H2D of the fp32 rows + mean / centre / fp16 shadow / sample"),
                     ("search_and_merge_s", f"search ({a.queries} x top-{a.top_k}, whole index on this GPU) + result download"), ("run_file_s", "run file")):
        lines.append(f"  {label:<72s} {tr.get(k, float('nan')):9.2f} s")
    lines.append(f"  search statistics: {getattr(RTP.main, 'last_search_stats', None)}")
    nlines = sum(1 for _ in open(out))
    lines.append(f"  run file: {os.path.getsize(out) / 1e6:.1f} MB, {nlines} lines")
    lines.append("")
    lines.append(f"whole job (cache excluded): {t_index + t_ret:.1f} s = {(t_index + t_ret) / 60:.1f} min; wall incl. synthetic cache and model setup {time.perf_counter() - t_job:.1f} s")
    lines.append("reference README.md:20: index step '~2.5h' on its hardware (context only: other GPUs, real tokeniser in the loop)")
    for ln in lines[-11:]:
        print(ln, flush=True)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as fh:
            fh.write("\n".join(lines) + "\n")
    shutil.rmtree(a.workdir, ignore_errors=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
