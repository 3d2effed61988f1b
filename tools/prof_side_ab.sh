# Per-kernel average durations of the training step with the query tower on its own stream (default) and in front of the passage tower
# (CLDRD_Q_SIDE=0): what the side stream's small launches cost the large GEMMs they run next to.
cd /tmp && export TMPDIR=/tmp
for side in 1 0; do
  export CLDRD_Q_SIDE=$side
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_side$side -o r -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-index --no-retrieve --no-kernel-events --no-ragged --no-ddp1 --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/prof_side$side.log 2>&1
  tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_side$side.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('Q_SIDE=$side', d['value'], d['ms_per_step'])"
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
def load(side):
    f = glob.glob(f'gpurun_out/prof_side{side}/**/*kernel_stats.csv', recursive=True)[0]
    return {r['Name']: (int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6) for r in csv.DictReader(open(f))}
a, b = load(1), load(0)
print(f"{'kernel':80s} {'calls':>6s} {'side us':>9s} {'inline us':>9s} {'d total ms/step':>14s}")
tot = 0.0
for k, (c, avg, t) in sorted(a.items(), key=lambda kv: -kv[1][2])[:24]:
    if k in b:
        d = (t - b[k][2]) / 25
        tot += d
        print(f"{k[:80]:80s} {c:6d} {avg:9.1f} {b[k][1]:9.1f} {d:14.3f}")
print("sum of the differences (ms/step):", round(tot, 3))
PY
