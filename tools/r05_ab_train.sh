# same-box A/B of library builds on the TRAIN leg only, many short rounds (the box drifts by ~0.3 % between rounds: interleave):
# bash tools/r05_ab_train.sh <rounds> <libA path rel. to repo> <libB> ...
cd $GRAFT_REPO_ROOT
rounds=$1; shift
for r in $(seq 1 $rounds); do for lib in "$@"; do
  out=$(CLDRD_LIB=$GRAFT_REPO_ROOT/$lib python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-index --no-retrieve --no-kernel-events --no-ddp1 --no-bf16-leg --no-ragged --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
  echo "[$r] $lib: $out"
done; done
