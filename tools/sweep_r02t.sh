#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { printf "%-88s" "$*"; timeout -k 10 200 python bench.py --warmup 8 --no-cpu-baseline --no-alone "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Mrays/s %.3f ms' % (d['value'], d['ms_per_step']))"; }
B="--opt closest_blocks_per_cu=8 --opt shadow_blocks_per_cu=8"
for rep in 1 2; do
run --config 1 --steps 64
run --config 1 --steps 64 --opt closest_variant=0 --opt closest_blocks_per_cu=8
run --config 1 --steps 64 --opt trace_variant=0 $B
run --config 2 --steps 32
run --config 2 --steps 32 --opt closest_variant=0 --opt closest_blocks_per_cu=8
run --config 2 --steps 32 --opt trace_variant=0 $B
done
