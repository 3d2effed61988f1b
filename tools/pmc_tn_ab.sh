# FETCH_SIZE of the weight-gradient group launch of the training step under environment variants; usage: bash tools/pmc_tn_ab.sh "A=1" "B=2" ...
cd /tmp && export TMPDIR=/tmp
i=0
for v in "$@"; do
  i=$((i+1))
  ( export $v; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_tn/$i -o r -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-index --no-retrieve --no-kernel-events --no-ragged --no-ddp1 --no-pmc --steps 3 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/pmc_tn_$i.log 2>&1 )
  python3 - "$GRAFT_REPO_ROOT/gpurun_out/pmc_tn/$i" "$v" <<'PY'
import csv, glob, sys
tot = {}
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'FETCH_SIZE' and ('gemm_tn' in r['Kernel_Name']):
            tot.setdefault(r['Dispatch_Id'], 0.0)
            tot[r['Dispatch_Id']] += float(r['Counter_Value'])
v = sorted(2.0 * x * 1024 / 1e9 for x in tot.values())
print(sys.argv[2], "TN launches:", len(v), "read GB of the large ones:", [round(x, 2) for x in v if x > 1.0])
PY
done
