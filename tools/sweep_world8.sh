#!/bin/bash
# a rank's share of a 4- / 8-rank tile partition on one GPU: frames per wavefront and frames in flight
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { printf "%-96s" "$*"; timeout -k 10 200 python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-alone "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Mrays/s %.3f ms' % (d['value'], d['ms_per_step']))"; }
B="--opt closest_blocks_per_cu=8 --opt shadow_blocks_per_cu=8"
for w in 8 4; do
run --emulate-world $w
run --emulate-world $w --opt trace_variant=0 $B
run --emulate-world $w --opt batch_frames=8
run --emulate-world $w --opt batch_frames=8 --opt trace_variant=0 $B
run --emulate-world $w --opt frames_in_flight=8
run --emulate-world $w --opt batch_frames=16 --opt frames_in_flight=2
done
