#!/bin/bash
# persistent-grid sizes of the traversal kernels: fewer blocks per CU leave LDS for the bandwidth-bound shading kernels to
# run on the same CUs at the same time (closest 25.6 KB, shadow 29.7 KB, shade_hit 20 KB per block; 160 KB per CU)
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
run() { printf "%-62s" "$*"; timeout -k 10 200 python bench.py --warmup 8 --no-cpu-baseline --no-alone "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Mrays/s %.3f ms' % (d['value'], d['ms_per_step']))"; }
for rep in 1 2; do
for cs in "6 5" "5 5" "5 4" "4 4" "4 3" "3 3" "6 3" "3 5"; do set -- $cs; run --steps 64 --opt closest_blocks_per_cu=$1 --opt shadow_blocks_per_cu=$2; done
done
for cs in "6 5" "5 4" "4 4" "4 3"; do set -- $cs; run --config 2 --steps 32 --opt closest_blocks_per_cu=$1 --opt shadow_blocks_per_cu=$2; run --config 3 --width 3840 --height 2160 --steps 16 --opt closest_blocks_per_cu=$1 --opt shadow_blocks_per_cu=$2; done
