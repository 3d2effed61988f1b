# same-box A/B of two library builds on the train AND index legs: bash tools/ab_lib_full.sh <libA> <libB>   (two rounds)
cd $GRAFT_REPO_ROOT
for r in 1 2; do for lib in "$@"; do
  out=$(CLDRD_LIB=$GRAFT_REPO_ROOT/cl-drd_amd/$lib python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-retrieve --no-kernel-events --no-ddp1 --no-bf16-leg --no-ragged --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'index', d.get('index', {}).get('value'))")
  echo "[$r] $lib: $out"
done; done
