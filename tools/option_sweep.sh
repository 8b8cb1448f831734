#!/bin/bash
# occupancy / overlap options against frame time (defaults first)
run() { printf "%-50s" "$*"; timeout -k 10 200 python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-alone "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Mrays/s %.3f ms' % (d['value'], d['ms_per_step']))"; }
run
for b in 6 7; do run --opt closest_blocks_per_cu=$b; done
for b in 4 5 7 8; do run --opt shadow_blocks_per_cu=$b; done
for f in 2 4; do run --opt frames_in_flight=$f; done
for b in 2 3 6 8; do run --opt batch_frames=$b; done
for v in 0 1 2 4; do run --opt shadow_variant=$v; done
