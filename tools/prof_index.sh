# rocprofv3 --kernel-trace --stats of the index (encode) leg of bench.py: forward of 512 passages x 128 tokens, 12 batches.
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_idx -o r -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-retrieve --no-kernel-events --no-ddp1 --no-ragged --no-pmc --steps 1 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/prof_idx.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_idx/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# keep the dispatches after the last adamw kernel (= the index leg)
last = max(i for i, r in enumerate(rows) if 'adamw' in r['Kernel_Name'])
rows = rows[last + 1:]
import collections
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    k = r['Kernel_Name'][:80]
    agg[k][0] += 1; agg[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
tot = sum(v[1] for v in agg.values())
print(f"index leg: {len(rows)} dispatches, kernel time {tot/1e6:.2f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"{k:80s} {v[0]:5d} {v[1]/1e6:8.2f} ms {v[1]/v[0]/1e3:8.1f} us {100*v[1]/tot:5.1f}%")
PY
