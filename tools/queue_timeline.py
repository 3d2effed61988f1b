#!/usr/bin/env python3
"""Per hardware queue: busy / idle spans of one replayed step (after tools/prof_train_amp.sh).  Prints idle gaps > 30 us per queue and the
kernel that ended them."""
import csv, glob, re, collections
f = glob.glob('gpurun_out/prof_amp/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])[:46], r['Queue_Id']) for r in rows)
ad = [i for i, e in enumerate(ev) if 'adamw' in e[2]]
seg = ev[ad[-3]:ad[-2] + 1]
t0 = seg[0][0]
print(f"step {(seg[-1][1] - t0) / 1e3:.1f} us")
last = {}
busy = collections.Counter()
for e in seg:
    q = e[3]
    busy[q] += e[1] - e[0]
    if q in last and e[0] - last[q][0] > 30000:
        print(f"q{q}: idle {(e[0] - last[q][0]) / 1e3:7.1f} us from {(last[q][0] - t0) / 1e3:8.1f} (after {last[q][1]}) until {e[2]}")
    last[q] = (e[1], e[2])
print({q: round(v / 1e3, 1) for q, v in busy.items()})
