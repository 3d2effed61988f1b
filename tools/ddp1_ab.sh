# the data-parallel code path with ONE rank over RCCL (bench.py --ddp1-child) under environment variants; usage: bash tools/ddp1_ab.sh "A=1" "B=2" ...
run() { env GPU_MAX_HW_QUEUES=${HWQ:-8} MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29500 + RANDOM % 2000)) RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 CLDRD_FORCE_DDP=1 $1 python bench.py --ddp1-child --steps 20 --warmup 6 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['samples_per_s'], d['ms_per_step'])"; }
for rep in 1 2; do
  for v in "$@"; do echo -n "[$rep] ${v:-default}: "; run "${v:-X=1}"; done
done
