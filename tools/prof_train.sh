# rocprofv3 --kernel-trace --stats of the train leg of bench.py (25 steps), summarised per kernel.
# Outputs under gpurun_out/prof2/: r_kernel_stats.csv (raw) and summary.txt (copy both to profiles/).
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof2 -o r -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-index --no-retrieve --no-kernel-events --no-ragged --no-ddp1 --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/prof2.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof2/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
steps = 25
out = ["# rocprofv3 --kernel-trace --stats of `python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-index --no-retrieve --no-kernel-events --no-ragged --no-ddp1`",
       "# (25 training steps of cfg2: DistilBERT-6L x2 towers, B=8, N=32, L=128, kl_div, dropout 0.1; 1x MI355X)",
       f"# total kernel time {tot/1e6:.1f} ms over {steps} steps = {tot/1e6/steps:.2f} ms/step (two streams overlap: wall time per step is lower)",
       "", f"{'kernel':86s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s} {'share':>6s}"]
for r in rows[:40]:
    out.append(f"{r['Name'][:86]:86s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['AverageNs'])/1e3:9.1f} {100*float(r['TotalDurationNs'])/tot:5.1f}%")
open('gpurun_out/prof2/summary.txt', 'w').write("\n".join(out) + "\n")
print("\n".join(out[:34]))
PY
