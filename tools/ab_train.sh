# same-box A/B of the train leg of bench.py under environment variants (samples/s, ms/step); usage: bash tools/ab_train.sh "A=1" "B=2 C=3" ...
run() { env $1 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-index --no-retrieve --no-kernel-events --no-ddp1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  for v in "$@"; do echo -n "[$rep] ${v:-default}: "; run "${v:-X=1}"; done
done
