# HBM traffic of the train step's kernels: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over the same
# bench command, aggregated per kernel.  FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md
# section HBM); both counters are in KiB.  Output: gpurun_out/pmc_train/traffic.json (copy to profiles/).
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_train/$c -o r -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-index --no-retrieve --no-kernel-events --no-ragged --no-ddp1 --steps 4 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/pmc_train_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f'gpurun_out/pmc_train/{c}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != c: continue
            k = r['Kernel_Name']
            agg[k][c] += float(r['Counter_Value']); calls[(k, c)].add(r['Dispatch_Id'])
out = {}
for k, v in agg.items():
    n = max(len(calls[(k, 'FETCH_SIZE')]), len(calls[(k, 'WRITE_SIZE')]), 1)
    rd = 2.0 * v.get('FETCH_SIZE', 0.0) * 1024 / max(len(calls[(k, 'FETCH_SIZE')]), 1)
    wr = v.get('WRITE_SIZE', 0.0) * 1024 / max(len(calls[(k, 'WRITE_SIZE')]), 1)
    out[k] = {"launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr}
json.dump(out, open('gpurun_out/pmc_train/traffic.json', 'w'), indent=1)
rows = sorted(out.items(), key=lambda kv: -(kv[1]["read_bytes_per_launch"] + kv[1]["write_bytes_per_launch"]) * kv[1]["launches"])
for k, v in rows[:24]:
    print(f"{k[:84]:84s} {v['launches']:5d} rd {v['read_bytes_per_launch']/1e6:9.2f} MB wr {v['write_bytes_per_launch']/1e6:9.2f} MB")
PY
