#!/usr/bin/env python3
"""Summary table of a tools/pmc_step.sh run: python3 tools/pmc_summary.py <output directory of the passes>  (-> summary.md, counters.json)"""
import csv, glob, collections, json, re, sys
out_dir = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(set))
dur_ns = collections.defaultdict(float)          # wall time of the dispatches GRBM_GUI_ACTIVE was read on (that pass's own timestamps)
def short(k):
    k = re.sub(r'\(anonymous namespace\)::', '', k)
    k = re.sub(r'^void ', '', k)
    return re.sub(r'\(.*$', '', k)[:64]
for f in glob.glob(out_dir + '/g*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r['Kernel_Name'])
        c = r['Counter_Name']
        agg[k][c] += float(r['Counter_Value'])
        cnt[k][c].add(r['Dispatch_Id'])
        if c == 'GRBM_GUI_ACTIVE' and r.get('End_Timestamp'):
            dur_ns[k] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
rows = {}
for k, v in agg.items():
    per = {c: v[c] / max(len(cnt[k][c]), 1) for c in v}
    n = max(len(s) for s in cnt[k].values())
    d = {"launches_seen": n}
    cyc = per.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
    if cyc > 0:
        d["cycles"] = round(cyc)
        d["mfma_busy"] = round(per.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / 1024.0 / cyc, 4)
        d["sq_busy"] = round(per.get('SQ_BUSY_CYCLES', 0.0) / 8.0 / cyc, 3)      # SQ_BUSY_CYCLES: per XCD as well
        d["insts_mfma"] = round(per.get('SQ_INSTS_MFMA', 0.0))
        if dur_ns[k] > 0:
            # the clock the chip held during these dispatches (MI355X_MICROARCH.md, DVFS give-back: GRBM_GUI_ACTIVE / 8 / wall time; reads high on
            # dispatches shorter than ~0.3 ms) and the dispatch time in the serialised counter pass
            d["us"] = round(dur_ns[k] / len(cnt[k]['GRBM_GUI_ACTIVE']) / 1e3, 1)
            d["clock_ghz"] = round(v['GRBM_GUI_ACTIVE'] / 8.0 / dur_ns[k], 2)
    wc = per.get('SQ_WAVE_CYCLES', 0.0)
    # groups 2 and 3 are different passes: ratios of one pass's counters to another's SQ_WAVE_CYCLES are good to the run-to-run spread
    if wc > 0:
        for c, name in (('SQ_WAIT_ANY', 'wait_any'), ('SQ_WAIT_INST_ANY', 'wait_inst_any'), ('SQ_WAIT_INST_LDS', 'wait_inst_lds'),
                        ('SQ_ACTIVE_INST_ANY', 'active_any'), ('SQ_ACTIVE_INST_VALU', 'active_valu'), ('SQ_ACTIVE_INST_LDS', 'active_lds')):
            if c in per: d[name] = round(per[c] / wc, 3)
        d["insts_valu"] = round(per.get('SQ_INSTS_VALU', 0.0)); d["insts_vmem"] = round(per.get('SQ_INSTS_VMEM', 0.0))
    if 'SQ_INSTS_LDS' in per:
        d["insts_lds"] = round(per['SQ_INSTS_LDS'])
        if per.get('SQ_LDS_IDX_ACTIVE', 0) > 0: d["lds_conflict_frac"] = round(per.get('SQ_LDS_BANK_CONFLICT', 0.0) / per['SQ_LDS_IDX_ACTIVE'], 4)
    if 'FETCH_SIZE' in per: d["read_MB"] = round(2.0 * per['FETCH_SIZE'] * 1024 / 1e6, 2)
    if 'WRITE_SIZE' in per: d["write_MB"] = round(per['WRITE_SIZE'] * 1024 / 1e6, 2)
    rows[k] = d
json.dump(rows, open(out_dir + '/counters.json', 'w'), indent=1)
keys = ["launches_seen", "us", "cycles", "clock_ghz", "mfma_busy", "wait_any", "wait_inst_any", "wait_inst_lds", "active_any", "active_valu", "active_lds",
        "lds_conflict_frac", "insts_mfma", "insts_valu", "insts_lds", "insts_vmem", "read_MB", "write_MB"]
order = sorted(rows, key=lambda k: -(rows[k].get("cycles", 0) * rows[k]["launches_seen"]))
lines = ["# per-launch averages, rocprofv3 --pmc (one pass per counter group) over `bench.py --steps 3 --warmup 4` (train leg only, cfg2)",
         "# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x GRBM_GUI_ACTIVE / 8); wait_* and active_* = counter / SQ_WAVE_CYCLES; read = 2 x FETCH_SIZE",
         "# us = dispatch time in the (serialised) counter pass; clock_ghz = cycles / us: the clock the chip HELD (2.4 GHz nominal; the quotient reads high below ~0.3 ms)", "",
         "| kernel | " + " | ".join(keys) + " |", "|---|" + "---|" * len(keys)]
for k in order[:36]:
    lines.append("| " + k + " | " + " | ".join(str(rows[k].get(x, "")) for x in keys) + " |")
open(out_dir + '/summary.md', 'w').write("\n".join(lines) + "\n")
print("\n".join(lines[:30]))