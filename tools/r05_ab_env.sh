# same-box A/B of environment settings on the train leg: bash tools/r05_ab_env.sh "X=1" "CLDRD_FOO=0" ...   (three rounds)
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for v in "$@"; do
  out=$(env $v python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-index --no-retrieve --no-kernel-events --no-ddp1 --no-bf16-leg --no-ragged --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['final_loss'])")
  echo "[$r] $v: $out"
done; done
