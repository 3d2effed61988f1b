# One GPU-box visit that produces a round's evidence under gpurun_out/<tag>/: the default bench line, the kernel stats of the training step,
# the per-kernel PMC table (tools/pmc_step.sh), the index / retrieve leg profiles (tools/prof_legs.sh), the packed index encode and the training
# command line's loop (tools/prof_index_ragged.sh, tools/prof_train_cli.sh, tools/time_train_cli.py).  Usage: bash tools/round_evidence.sh r06
TAG=${1:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 bench.py > $OUT/bench_line.json 2> $OUT/bench.err
tail -c 300 $OUT/bench.err
bash tools/prof_train_amp.sh > $OUT/kernel_stats_train.txt 2>&1
bash tools/pmc_step.sh $TAG > $OUT/pmc_step.log 2>&1
cp gpurun_out/pmc_step_$TAG/summary.md $OUT/pmc_step_summary.md 2>/dev/null
cp gpurun_out/pmc_step_$TAG/counters.json $OUT/pmc_counters.json 2>/dev/null
python3 tools/traffic_json.py gpurun_out/pmc_step_$TAG/counters.json 7 > $OUT/train_step_hbm_traffic.json 2>/dev/null
bash tools/prof_legs.sh $TAG > $OUT/prof_legs.txt 2>&1
cp gpurun_out/${TAG}_index_leg_summary.txt gpurun_out/${TAG}_retrieve_summary.txt $OUT/ 2>/dev/null
bash tools/prof_index_ragged.sh $TAG > $OUT/prof_index_packed.txt 2>&1
bash tools/prof_train_cli.sh $TAG > $OUT/prof_train_cli.txt 2>&1
cp gpurun_out/${TAG}_index_packed_summary.txt gpurun_out/${TAG}_train_cli_summary.txt $OUT/ 2>/dev/null
(python3 tools/time_train_cli.py 400; python3 tools/time_train_cli.py 400 --fixed) 2>/dev/null | grep "trainer CLI" > $OUT/train_cli_timing.txt
python3 -c "
import json
d = json.loads(open('$OUT/bench_line.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step', 'dtype', 'step_launch', 'step_mfma_frac')}, d.get('bf16_operand_mode'), d['roofline']['frac'], d['roofline']['traffic'], d['index'], d['retrieve'].get('path_hbm_frac'), d['retrieve'].get('merge8_ms'))
"
