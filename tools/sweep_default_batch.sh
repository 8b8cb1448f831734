#!/bin/bash
# frames per wavefront under the default bench.py invocation (warm-up 4, 64 / 32 / 16 steps)
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { printf "%-50s" "$*"; timeout -k 10 300 python bench.py --no-cpu-baseline --no-alone "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Mrays/s %.3f ms' % (d['value'], d['ms_per_step']))"; }
for rep in 1 2 3; do for b in 4 8 16; do run --opt batch_frames=$b; done; done
for b in 4 8 16; do run --steps 32 --opt batch_frames=$b; run --steps 16 --opt batch_frames=$b; done
