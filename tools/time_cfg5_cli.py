#!/usr/bin/env python3
"""cfg5 end to end through the command line of the retrieve path: a 1 105 228-row fp32 shard (8 841 823 / 8) written
as an index file, then `retrieve_top_passages.main` with 6 980 synthetic queries, k = 1000: seconds of every phase
(model load / query encode / index read / H2D + shadows / search + merge / run file).

    python tools/time_cfg5_cli.py [rows] [queries]
"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cldrd_amd.encoder import EncoderConfig
from cldrd_amd.models import NwayDualEncoder
from cldrd_amd.retriever import retrieve_top_passages as RTP
from cldrd_amd.retriever import retrieval_utils as RU

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1105228
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 6980
with tempfile.TemporaryDirectory(dir="/tmp") as td:
    torch.manual_seed(0)
    cfg = EncoderConfig(arch="distilbert")
    model = NwayDualEncoder(cfg, share_weights=False)
    mdir = os.path.join(td, "model")
    model.query_encoder.save_pretrained(mdir)
    ckpt = os.path.join(td, "checkpoint_1.pth.tar")
    torch.save({"state_dict": {"module." + k: v for k, v in model.state_dict().items()}}, ckpt)
    del model
    # index rows in the direction of what the RANDOM-INIT query tower produces would need an encode of the whole shard; the search cost does
    # not depend on it, so the shard is the bench corpus (unit Gaussian direction x U(9, 12) norm), written through the package's writer
    t0 = time.perf_counter()
    g = torch.Generator(device="cuda").manual_seed(1234)
    P = torch.randn(rows, 768, device="cuda", generator=g)
    P *= (9.0 + 3.0 * torch.rand(rows, 1, device="cuda", generator=g)) / P.norm(dim=1, keepdim=True)
    idx = RU.construct_flatindex_from_embeddings(P.cpu().numpy(), np.arange(rows, dtype=np.int64))
    del P
    torch.cuda.empty_cache()
    ipath = os.path.join(td, "checkpoint_1.index")
    RU.write_index(idx, ipath)
    del idx
    print(f"setup (generate + write a {rows} x 768 fp32 index file): {time.perf_counter() - t0:.1f} s", flush=True)
    out = os.path.join(td, "runs", "dev.run")
    args = RTP.get_args(["--resume", ckpt, "--model_name_or_path", mdir, "--index_path", ipath, "--max_length", "30", "--top_k", "1000",
                         "--synthetic_queries", str(nq), "--output_path", out])
    t0 = time.perf_counter()
    RTP.main(args)
    wall = time.perf_counter() - t0
    tm = RTP.main.last_timings
    print("cfg5 CLI phases (s): " + ", ".join(f"{k[:-2]} {v:.3f}" for k, v in tm.items()) + f" | total {wall:.3f}")
    print(f"run file: {os.path.getsize(out) / 1e6:.1f} MB, {sum(1 for _ in open(out))} lines")
