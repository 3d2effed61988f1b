# rocprofv3 --kernel-trace --stats of the PACKED index encode (512 MS MARCO-shaped passages per batch, padded to the longest of the batch as the
# tokenizer pads them, token counts given) at max_length 128 and 256; summary -> gpurun_out/<tag>_index_packed_summary.txt
export TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for L in 128 256; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_idxr$L -o r -- python3 $R/tools/index_prof.py $L 10 ragged > $R/gpurun_out/prof_idxr$L.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, os
TAG = os.environ.get('TAG', 'r06')
out = []
for L in (128, 256):
    f = glob.glob(f'gpurun_out/prof_idxr{L}/**/*kernel_stats.csv', recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    log = [l for l in open(f'gpurun_out/prof_idxr{L}.log').read().splitlines() if 'passages/s' in l][-1]
    out += [f"# rocprofv3 --kernel-trace --stats -- python3 tools/index_prof.py {L} 10 ragged: 12 forward passes (2 warm-up + 10) of 512 MS MARCO-shaped passages, packed,",
            f"# DistilBERT-6L passage tower, evaluation mode; under the profiler: {log}", f"# total kernel time {tot/1e6:.1f} ms = {tot/1e6/12:.3f} ms per batch of 512", "",
            f"{'kernel':92s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s} {'share':>6s}"]
    for r in rows[:16]:
        out.append(f"{r['Name'][:92]:92s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['AverageNs'])/1e3:9.1f} {100*float(r['TotalDurationNs'])/tot:5.1f}%")
    out.append("")
open(f'gpurun_out/{TAG}_index_packed_summary.txt', 'w').write("\n".join(out) + "\n")
print("\n".join(out))
PY
