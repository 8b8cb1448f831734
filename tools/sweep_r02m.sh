#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { printf "%-72s" "$*"; timeout -k 10 200 python bench.py --warmup 8 --no-cpu-baseline --no-alone "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.1f Mrays/s %.3f ms | closest %.3f ms/launch shadow_ms %.1f shade_ms %.1f' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['trace_shadow_ms'], r['shade_ms']))"; }
run --config 2 --steps 32 --opt stream_nt=1
run --config 2 --steps 32 --opt stream_nt=1 --opt trace_variant=0 --opt closest_blocks_per_cu=8 --opt shadow_blocks_per_cu=8
run --config 2 --steps 32 --opt stream_nt=1 --opt trace_variant=2
run --config 2 --steps 32 --opt stream_nt=1 --opt frames_in_flight=6
run --config 2 --steps 32 --opt stream_nt=1 --opt frames_in_flight=8
run --config 1 --steps 32 --opt stream_nt=1 --opt batch_frames=1
run --config 1 --steps 32 --opt stream_nt=1 --opt batch_frames=1 --opt trace_variant=0 --opt closest_blocks_per_cu=8 --opt shadow_blocks_per_cu=8
run --config 1 --steps 32 --opt stream_nt=1 --opt batch_frames=1 --opt frames_in_flight=8
