# rocprofv3 --kernel-trace --stats of the index leg (L = 128 and L = 256) and of the retrieve leg; summaries -> gpurun_out/<tag>_*_summary.txt
export TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for L in 128 256; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_idx$L -o r -- python3 $R/tools/index_prof.py $L 10 > $R/gpurun_out/prof_idx$L.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ret -o r -- python3 $R/tools/retrieve_prof.py > $R/gpurun_out/prof_ret.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, os
TAG = os.environ.get('TAG', 'r06')
out = []
for L in (128, 256):
    f = glob.glob(f'gpurun_out/prof_idx{L}/**/*kernel_stats.csv', recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    log = [l for l in open(f'gpurun_out/prof_idx{L}.log').read().splitlines() if 'passages/s' in l][-1]
    out += [f"# rocprofv3 --kernel-trace --stats -- python3 tools/index_prof.py {L} 10: 12 forward passes (2 warm-up + 10) of 512 passages x {L} tokens, DistilBERT-6L",
            f"# passage tower, evaluation mode (bench.py index leg{' l256' if L == 256 else ''}); under the profiler: {log}",
            f"# total kernel time {tot/1e6:.1f} ms = {tot/1e6/12:.3f} ms per batch of 512", "",
            f"{'kernel':92s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s} {'share':>6s}"]
    for r in rows[:18]:
        out.append(f"{r['Name'][:92]:92s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['AverageNs'])/1e3:9.1f} {100*float(r['TotalDurationNs'])/tot:5.1f}%")
    out.append("")
open(f'gpurun_out/{TAG}_index_leg_summary.txt', 'w').write("\n".join(out) + "\n")
print("\n".join(out))
f = glob.glob('gpurun_out/prof_ret/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
out = ["# rocprofv3 --kernel-trace --stats -- python3 tools/retrieve_prof.py: 3 device-resident searches of 6980 queries (165 batches of 128), k = 1000,",
       "# one cfg5 shard (1 105 228 x 768 rows, fp32 + fp16 shadow), 1x MI355X", f"# total kernel time {tot/1e6:.1f} ms = {tot/1e6/165:.3f} ms per 128-query batch",
       "", f"{'kernel':92s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s} {'share':>6s}"]
for r in rows[:16]:
    out.append(f"{r['Name'][:92]:92s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['AverageNs'])/1e3:9.1f} {100*float(r['TotalDurationNs'])/tot:5.1f}%")
open(f'gpurun_out/{TAG}_retrieve_summary.txt', 'w').write("\n".join(out) + "\n")
print("\n".join(out))
PY
tail -1 gpurun_out/prof_ret.log
