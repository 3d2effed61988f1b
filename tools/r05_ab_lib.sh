# same-box A/B of two library builds on the train AND index legs: bash tools/r05_ab_lib.sh <libA path rel. to repo> <libB> ...   (two rounds)
cd $GRAFT_REPO_ROOT
for r in 1 2; do for lib in "$@"; do
  out=$(CLDRD_LIB=$GRAFT_REPO_ROOT/$lib python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-retrieve --no-kernel-events --no-ddp1 --no-bf16-leg --no-ragged --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'index', d['index']['value'], d['index']['mfma_frac'], 'l256', d['index']['l256']['value'], d['index']['l256']['mfma_frac'])")
  echo "[$r] $lib: $out"
done; done
