cd /tmp && export TMPDIR=/tmp
export ATTN_ITERS=1
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_attn/$tag -o r -- python3 $GRAFT_REPO_ROOT/tools/attn_bench.py > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_attn/*/**/*counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        if 'attn' not in k: continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
    for k, v in agg.items():
        print(k, {c: round(x / cnt[(k, c)]) for c, x in v.items()})
PY
