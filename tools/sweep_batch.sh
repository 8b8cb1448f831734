#!/bin/bash
# frames per wavefront (option batch_frames) by workload: 256 x 256, 4K, the iso-surface scene, config 1
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { printf "%-80s" "$*"; timeout -k 10 300 python bench.py --warmup 16 --no-cpu-baseline --no-alone "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Mrays/s %.3f ms' % (d['value'], d['ms_per_step']))"; }
for b in 0 8 16; do run --config 0 --width 256 --height 256 --steps 128 --opt batch_frames=$b; done
for b in 1 2 4; do run --config 3 --width 3840 --height 2160 --steps 16 --opt batch_frames=$b; done
for b in 2 4 8; do run --config 4 --steps 64 --opt batch_frames=$b; done
for b in 2 4 8; do run --steps 64 --opt batch_frames=$b; done
