cd /tmp && export TMPDIR=/tmp
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_gemm/$tag -o r -- python3 $GRAFT_REPO_ROOT/tools/gemm_pmc.py > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
res = collections.defaultdict(dict)
for f in sorted(glob.glob('gpurun_out/pmc_gemm/*/**/*counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'gemm' not in k: continue
        key = k[:70] + ' grid=' + r.get('Grid_Size', '?')
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        for c, xs in v.items():
            res[k][c] = xs[-1]
for k, v in res.items():
    print(k)
    print('   ', {c: int(x) for c, x in v.items()})
    if 'SQ_WAVE_CYCLES' in v and 'SQ_WAIT_ANY' in v:
        wc = v['SQ_WAVE_CYCLES']
        print(f"    wait_any {v['SQ_WAIT_ANY']/wc:.2f}  wait_inst_any {v.get('SQ_WAIT_INST_ANY',0)/wc:.2f}  active_any {v.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f}  valu {v.get('SQ_ACTIVE_INST_VALU',0)/wc:.2f}  lds {v.get('SQ_ACTIVE_INST_LDS',0)/wc:.2f}")
    if 'GRBM_GUI_ACTIVE' in v:
        cyc = v['GRBM_GUI_ACTIVE'] / 8
        print(f"    cycles {cyc:.0f}  mfma_busy {v['SQ_VALU_MFMA_BUSY_CYCLES']/1024/cyc:.2f}")
PY
