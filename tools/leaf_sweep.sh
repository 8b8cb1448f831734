#!/bin/bash
# builder leaf parameters (max triangles per leaf, SAH traversal cost x100) against frame time
run() { printf "%-60s" "$*"; timeout -k 10 200 python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-alone "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.1f Mrays/s %.3f ms  nodes/ray %.2f tris/ray %.2f' % (d['value'], d['ms_per_step'], r['nodes_per_ray'], r['tris_per_ray']))"; }
run
for l in 1 2 3 6 8; do run --opt bvh_max_leaf=$l; done
for c in 25 100 150 250; do run --opt bvh_sah_cost_x100=$c; done
run --opt bvh_max_leaf=8 --opt bvh_sah_cost_x100=100
run --opt bvh_max_leaf=2 --opt bvh_sah_cost_x100=100
