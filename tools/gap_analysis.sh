# Where is the GPU idle inside a training step?  rocprofv3 --kernel-trace of the train leg, then per step (adamw to adamw):
# wall, union of the intervals in which ANY kernel runs, and the idle gaps by the kernel that follows them.
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_gap -o r -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-index --no-retrieve --no-kernel-events --no-pmc --no-ddp1 --no-ragged --no-bf16-leg --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/prof_gap.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/prof_gap/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '')) for r in rows)
ad = [i for i, e in enumerate(ev) if 'adamw' in e[2]]
print("queues:", collections.Counter(e[3] for e in ev))
tot_wall = tot_busy = 0
gaps = collections.defaultdict(lambda: [0, 0])
for a, b in zip(ad[4:-1], ad[5:]):
    seg = ev[a + 1:b + 1]
    t0, t1 = ev[a][1], ev[b][1]
    cur_end = t0
    busy = 0
    for s, e, name, q in seg:
        if s > cur_end:
            g = gaps[name[:70]]
            g[0] += 1; g[1] += s - cur_end
            busy += e - s
            cur_end = e
        elif e > cur_end:
            busy += e - cur_end
            cur_end = e
    tot_wall += t1 - t0; tot_busy += busy
n = len(ad[5:]) - 0
n = len(list(zip(ad[4:-1], ad[5:])))
print(f"steps analysed {n}: wall {tot_wall/n/1e6:.3f} ms/step, some kernel running {tot_busy/n/1e6:.3f} ms/step, idle {(tot_wall-tot_busy)/n/1e6:.3f} ms/step")
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"idle before {k:70s} {v[0]/n:6.1f}/step {v[1]/n/1e3:8.1f} us/step  avg {v[1]/v[0]/1e3:6.2f} us")
PY
