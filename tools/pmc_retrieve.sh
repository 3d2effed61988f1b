# HBM-side traffic of the retrieve leg's kernels: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/retrieve_prof.py
# (FETCH_SIZE doubled: gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md section HBM; both counters in KiB).
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_ret/$c -o r -- python3 $GRAFT_REPO_ROOT/tools/retrieve_prof.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_ret_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f'gpurun_out/pmc_ret/{c}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != c: continue
            k = r['Kernel_Name']
            agg[k][c] += float(r['Counter_Value']); calls[(k, c)].add(r['Dispatch_Id'])
out = {}
for k, v in agg.items():
    if not any(t in k for t in ("scan_stream", "rescore", "select_compact", "topk_sort", "kth_largest")): continue
    n = max(len(calls[(k, 'FETCH_SIZE')]), len(calls[(k, 'WRITE_SIZE')]), 1)
    rd = 2.0 * v.get('FETCH_SIZE', 0.0) * 1024 / max(len(calls[(k, 'FETCH_SIZE')]), 1)
    wr = v.get('WRITE_SIZE', 0.0) * 1024 / max(len(calls[(k, 'WRITE_SIZE')]), 1)
    out[k[:100]] = {"launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr}
json.dump(out, open('gpurun_out/pmc_ret/traffic.json', 'w'), indent=1)
for k, v in out.items():
    print(f"{k[:84]:84s} {v['launches']:5d} rd {v['read_bytes_per_launch']/1e6:9.2f} MB wr {v['write_bytes_per_launch']/1e6:9.2f} MB")
PY
