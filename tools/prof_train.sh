cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof2 -o r -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-index --no-retrieve --no-kernel-events --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/prof2.log 2>&1
cd $GRAFT_REPO_ROOT; ls gpurun_out/prof2 | head; python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof2/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:28]:
    print(f"{r['Name'][:90]:90s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:9.1f} us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
print('total ms', tot/1e6)
PY
