#!/bin/bash
# in-frame tuning of the refill kernels on the 48-byte-node tree: refill threshold (variants 2/3/4 = 4/8/16 idle lanes; 1 = chained steps), blocks per CU
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { printf "%-72s" "$*"; timeout -k 10 200 python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-alone "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.1f Mrays/s %.3f ms | closest %.3f ms/launch shadow_ms %.1f shade_ms %.1f' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['trace_shadow_ms'], r['shade_ms']))"; }
run
run --opt trace_variant=2
run --opt trace_variant=4
run --opt trace_variant=1
run --opt trace_variant=0 --opt closest_blocks_per_cu=8 --opt shadow_blocks_per_cu=8
run --opt closest_blocks_per_cu=5
run --opt closest_blocks_per_cu=5 --opt shadow_blocks_per_cu=4
run --opt closest_blocks_per_cu=4 --opt shadow_blocks_per_cu=4
run --opt frames_in_flight=3
run --opt frames_in_flight=6
run --opt batch_frames=2
run --opt batch_frames=8
