# rocprofv3 --kernel-trace --stats of the training command line's loop (tools/time_train_cli.py, loader in the main process so that nothing forks
# under the profiler): MS MARCO-shaped (packed, eager) and fixed-length (padded, graph replay) batches; summaries -> gpurun_out/<tag>_train_cli_summary.txt
export TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cli_ragged -o r -- python3 $R/tools/time_train_cli.py 120 --workers 0 > $R/gpurun_out/prof_cli_ragged.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cli_fixed -o r -- python3 $R/tools/time_train_cli.py 120 --workers 0 --fixed > $R/gpurun_out/prof_cli_fixed.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, os
TAG = os.environ.get('TAG', 'r06')
out = []
for kind in ("ragged", "fixed"):
    f = glob.glob(f'gpurun_out/prof_cli_{kind}/**/*kernel_stats.csv', recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    log = [l for l in open(f'gpurun_out/prof_cli_{kind}.log').read().splitlines() if 'samples/s' in l][-1]
    out += [f"# rocprofv3 --kernel-trace --stats -- python3 tools/time_train_cli.py 120 --workers 0{' --fixed' if kind == 'fixed' else ''}", f"# under the profiler: {log}",
            f"# total kernel time {tot/1e6:.1f} ms over 120 steps = {tot/1e6/120:.3f} ms of kernels per step (streams overlap: the sum exceeds the wall clock)", "",
            f"{'kernel':100s} {'calls':>7s} {'total_ms':>9s} {'avg_us':>9s} {'share':>6s} {'ms/step':>8s}"]
    for r in rows[:26]:
        out.append(f"{r['Name'][:100]:100s} {int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['AverageNs'])/1e3:9.1f} {100*float(r['TotalDurationNs'])/tot:5.1f}% {float(r['TotalDurationNs'])/1e6/120:8.3f}")
    out.append("")
open(f'gpurun_out/{TAG}_train_cli_summary.txt', 'w').write("\n".join(out) + "\n")
print("\n".join(out))
PY
