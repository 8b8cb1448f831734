#!/bin/bash
run() { printf "%-50s" "$*"; timeout -k 10 200 python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-alone "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Mrays/s %.3f ms' % (d['value'], d['ms_per_step']))"; }
run
for b in 4 6 7; do run --opt closest_blocks_per_cu=$b; done
for b in 4 5 7 8; do run --opt shadow_blocks_per_cu=$b; done
for f in 1 2 4 5; do run --opt frames_in_flight=$f; done
for b in 1 3 4; do run --opt batch_frames=$b; done
run --opt batch_frames=1 --opt frames_in_flight=6
run --opt batch_frames=3 --opt frames_in_flight=2
run --opt overlap_miss=0
run --opt overlap_shadow=0
