#!/usr/bin/env python3
"""Kernel trace of one replayed step between the passage tower's last forward FFN2 and its first large data-gradient GEMM: queue id, start,
duration of every kernel (run after tools/prof_train_amp.sh: reads gpurun_out/prof_amp)."""
import csv, glob, re
f = glob.glob('gpurun_out/prof_amp/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])[:46], r['Queue_Id']) for r in rows)
ad = [i for i, e in enumerate(ev) if 'adamw' in e[2]]
seg = ev[ad[-3]:ad[-2] + 1]
t0 = seg[0][0]
i0 = max(i for i, e in enumerate(seg) if '192, 2929' in e[2])
i1 = min(i for i, e in enumerate(seg) if '256, 1544' in e[2])
print(f"step {(seg[-1][1] - t0) / 1e3:.1f} us; last forward FFN2 ends {(seg[i0][1] - t0) / 1e3:.1f}, first large data-gradient GEMM starts {(seg[i1][0] - t0) / 1e3:.1f}: "
      f"{(seg[i1][0] - seg[i0][1]) / 1e3:.1f} us between them")
last = {}
for e in seg[i0:i1 + 1]:
    q = e[3]
    gap = (e[0] - last[q]) / 1e3 if q in last else 0.0
    last[q] = e[1]
    print(f"{(e[0] - t0) / 1e3:9.1f} {(e[1] - e[0]) / 1e3:7.1f} q{q} {'(idle %.0f us on this queue) ' % gap if gap > 20 else ''}{e[2]}")
