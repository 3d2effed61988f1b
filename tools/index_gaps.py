#!/usr/bin/env python3
"""GPU idle inside the index-encode leg (eager launches): after tools/prof_index.sh, wall span of the leg's dispatches against the union of
their execution intervals, and the largest gaps by the kernel that follows them."""
import csv, glob, collections, re
f = glob.glob('gpurun_out/prof_idx/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
last = max(i for i, r in enumerate(rows) if 'adamw' in r['Kernel_Name'])
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])[:50]) for r in rows[last + 1:])
# batches are separated by long gaps (host sync + timing code): split at gaps > 300 us
batches, cur = [], [ev[0]]
for a, b in zip(ev[:-1], ev[1:]):
    if b[0] - a[1] > 300000:
        batches.append(cur); cur = []
    cur.append(b)
batches.append(cur)
print("groups of back-to-back dispatches:", [len(b) for b in batches][:12])
gaps = collections.defaultdict(lambda: [0, 0])
wall = busy = 0
for b in batches:
    if len(b) < 40:
        continue
    wall += b[-1][1] - b[0][0]
    end = b[0][0]
    for s, e, n in b:
        if s > end:
            gaps[n][0] += 1; gaps[n][1] += s - end
        busy += max(0, e - max(s, end)); end = max(end, e)
print(f"wall {wall / 1e6:.2f} ms, kernels running {busy / 1e6:.2f} ms, idle {(wall - busy) / 1e6:.2f} ms = {100 * (wall - busy) / wall:.1f} %")
for n, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:10]:
    print(f"  idle before {n:50s} {v[0]:5d} x avg {v[1] / v[0] / 1e3:6.1f} us = {v[1] / 1e6:.2f} ms")
