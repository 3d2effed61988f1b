#!/usr/bin/env python3
"""profiles/r05_train_step_hbm_traffic.json from the per-kernel PMC pass of tools/pmc_step.sh (counters.json): memory-side bytes per launch
(reads = 2 x FETCH_SIZE KiB, gfx950 correction; writes = WRITE_SIZE) x launches per step, per kernel and in total.  Usage:
python tools/traffic_json.py gpurun_out/pmc_step_amp16/counters.json 7 > profiles/r05_train_step_hbm_traffic.json   (7 = steps the pass ran)"""
import json, sys
d = json.load(open(sys.argv[1]))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 7.0
rows, tot_r, tot_w = [], 0.0, 0.0
for k, v in d.items():
    n = v.get("launches_seen", 0) / steps
    r, w = v.get("read_MB") or 0.0, v.get("write_MB") or 0.0
    if n * (r + w) <= 0:
        continue
    rows.append({"kernel": k, "launches_per_step": round(n, 2), "read_MB_per_launch": round(r, 2), "write_MB_per_launch": round(w, 2),
                 "MB_per_step": round(n * (r + w), 1)})
    tot_r += n * r
    tot_w += n * w
rows.sort(key=lambda x: -x["MB_per_step"])
print(json.dumps({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) over bench.py's train leg, cfg2, all-fp16 mode",
                  "unit": "MB = 1e6 bytes at the memory side of L2 (Infinity Cache hits included)", "steps_in_pass": steps,
                  "total_GB_per_step": round((tot_r + tot_w) / 1e3, 2), "read_GB_per_step": round(tot_r / 1e3, 2), "write_GB_per_step": round(tot_w / 1e3, 2),
                  "kernels": rows}, indent=1))
