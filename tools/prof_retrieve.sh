# rocprofv3 --kernel-trace --stats of the retrieve leg (tools/retrieve_prof.py: 3 searches of 6980 queries, k = 1000, 1.1 M-row shard)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ret -o r -- python3 $GRAFT_REPO_ROOT/tools/retrieve_prof.py > $GRAFT_REPO_ROOT/gpurun_out/prof_ret.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_ret/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
out = ["# rocprofv3 --kernel-trace --stats of tools/retrieve_prof.py: 3 device-resident searches of 6980 queries (165 batches of 128), k = 1000,",
       "# one cfg5 shard (1 105 228 x 768 rows, fp32 + fp16 shadow), 1x MI355X", f"# total kernel time {tot/1e6:.1f} ms = {tot/1e6/165:.3f} ms per 128-query batch",
       "", f"{'kernel':90s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s} {'share':>6s}"]
for r in rows[:16]:
    out.append(f"{r['Name'][:90]:90s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['AverageNs'])/1e3:9.1f} {100*float(r['TotalDurationNs'])/tot:5.1f}%")
open('gpurun_out/prof_ret/summary.txt', 'w').write("\n".join(out) + "\n")
print("\n".join(out))
PY
tail -2 gpurun_out/prof_ret.log
