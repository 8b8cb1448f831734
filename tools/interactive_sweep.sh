#!/bin/bash
# frame latency for an interactive caller (a synchronisation after every frame) under the grid-size / overlap options
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { printf "%-70s" "$*"; timeout -k 10 300 python bench.py --no-cpu-baseline --steps 16 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms batched | serial %.3f | interactive %.3f' % (d['ms_per_step'], d['config']['frame_by_frame_ms'], d['config']['interactive_frame_ms']))"; }
run
run --opt trace_variant=0 --opt closest_blocks_per_cu=8 --opt shadow_blocks_per_cu=8
run --opt closest_blocks_per_cu=4 --opt shadow_blocks_per_cu=4
run --opt closest_blocks_per_cu=3 --opt shadow_blocks_per_cu=3
run --opt closest_blocks_per_cu=2 --opt shadow_blocks_per_cu=2
run --opt overlap_shadow=0
run --opt overlap_miss=0
