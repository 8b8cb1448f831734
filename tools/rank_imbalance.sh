#!/bin/bash
# load balance of the tile partition: every rank's share of an 8-rank job, one after the other on one GPU
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
for tile in 64 32; do
for r in 0 1 2 3 4 5 6 7; do
  printf "world 8 tile %d rank %d: " $tile $r; timeout -k 10 200 python bench.py --emulate-world 8 --emulate-rank $r --tile $tile --steps 64 --warmup 16 --no-cpu-baseline --no-alone 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms/frame  %.2f M rays/frame' % (d['ms_per_step'], d['config'].get('rays_per_frame', 0)/1e6))"
done
done
