# Per-kernel SQ / TCC counters of the TRAINING STEP (bench.py's train leg, cfg2), one rocprofv3 --pmc pass per counter group
# (never combined with the tracing domains; --kernel-trace only).  Usage on the GPU box:  bash tools/pmc_step.sh [tag]
# Output: gpurun_out/pmc_step_<tag>/summary.md (copy to profiles/) + counters.json.
#   mfma_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles), cycles = GRBM_GUI_ACTIVE / 8 (sum over the 8 XCDs)
#   wait_*      = SQ_WAIT_* / SQ_WAVE_CYCLES (quad-cycle units on both sides)
#   read / write = 2 x FETCH_SIZE / WRITE_SIZE (KiB; gfx950 tallies the 128-B requests of wide reads at 64 B: MI355X_MICROARCH.md, HBM)
TAG=${1:-base}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_step_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o r -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-index --no-retrieve --no-kernel-events --no-pmc --no-ragged --no-ddp1 --steps 3 --warmup 4 > $OUT/g$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $OUT
