# kernel trace of a few training steps; prints one step (AdamW to AdamW) around its boundaries: tools/step_trace.sh [first n / last n rows]
cd /tmp && export TMPDIR=/tmp
export CLDRD_GRAPH=${TRACE_GRAPH:-0}   # rocprofv3 serialises the branches of a replayed graph: trace the eager step
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_trace -o r -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-index --no-retrieve --no-kernel-events --no-ragged --no-ddp1 --steps 12 --warmup 6 > $GRAFT_REPO_ROOT/gpurun_out/prof_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob('gpurun_out/prof_trace/**/*kernel_trace.csv', recursive=True)[0])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '')) for r in rows)
ad = [i for i, e in enumerate(ev) if 'adamw' in e[2]]
a, b = ad[-3], ad[-2]
seg = ev[a:b + 1]
t0 = seg[0][1]
print("step wall (adamw end to adamw end): %.1f us, %d launches" % ((seg[-1][1] - seg[0][1]) / 1e3, len(seg) - 1))
import collections
qs = collections.Counter(q for _, _, _, q in seg)
mainq = max(qs, key=lambda q: sum(e - s for s, e, _, qq in seg if qq == q))
print("queues:", dict(qs), "main =", mainq)
# the main queue's timeline: kernels and the idle gaps between them
cur = t0; idle = 0
rows_main = [(s, e, n) for s, e, n, q in seg if q == mainq]
for s, e, n in rows_main:
    if s > cur: idle += s - cur
    cur = max(cur, e)
print("main queue: busy %.1f us, idle %.1f us" % ((sum(e - s for s, e, n in rows_main)) / 1e3, idle / 1e3))
k = 0
for i, (s, e, n, q) in enumerate(seg):
    if i < 30 or i > len(seg) - 14 or (q == mainq and k < 12):
        if q == mainq: k += 1
        short = n.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')[:64]
        print(f"{i:4d} q{q} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  {short}")
gaps = []
cur = t0
for s, e, n in rows_main:
    if s - cur > 3000: gaps.append(((s - cur) / 1e3, (s - t0) / 1e3, n[:50]))
    cur = max(cur, e)
print("main-queue gaps > 3 us:", [(round(g, 1), round(t, 0), n) for g, t, n in gaps][:30])
PY
