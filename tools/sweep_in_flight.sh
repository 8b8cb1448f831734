#!/bin/bash
# wavefronts in flight (option frames_in_flight) with the path-count batch rule
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { printf "%-60s" "$*"; timeout -k 10 300 python bench.py --no-cpu-baseline --no-alone "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Mrays/s %.3f ms' % (d['value'], d['ms_per_step']))"; }
for rep in 1 2; do for f in 4 3 2 1; do run --opt frames_in_flight=$f; done; done
for f in 4 3 2; do run --config 2 --steps 32 --opt frames_in_flight=$f; run --config 4 --opt frames_in_flight=$f; run --emulate-world 8 --opt frames_in_flight=$f; done
