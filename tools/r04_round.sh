# One GPU-box visit that produces the round's evidence under gpurun_out/r04/: the default bench line, the kernel stats of the training step,
# and the per-kernel PMC table (tools/pmc_step.sh).  Usage: bash tools/r04_round.sh
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 bench.py > $OUT/bench_line.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.err
bash tools/prof_train_amp.sh > $OUT/prof_train.log 2>&1
bash tools/pmc_step.sh amp16 > $OUT/pmc_step.log 2>&1
cp gpurun_out/pmc_step_amp16/summary.md $OUT/pmc_step_summary.md 2>/dev/null
python3 -c "
import json
d = json.loads(open('$OUT/bench_line.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step', 'dtype', 'step_launch')}, d.get('bf16_operand_mode'), d['roofline']['frac'], d.get('index_encode', d.get('index')), d['retrieve'].get('path_hbm_frac'), d['retrieve'].get('cls_like'))
"
