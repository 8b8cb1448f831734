#!/bin/bash
# frame rate against scene size (--detail scales the tessellation): how much of the frame time is the BVH working set missing the 4 MiB L2 of an XCD?
cd ${GRAFT_REPO_ROOT:-/root/repo}
for d in 1.0 0.5 0.25 0.1 0.03; do
  printf "detail %-5s " $d
  timeout -k 10 200 python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-alone --detail $d --tex-size 64 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%s | %.1f Mrays/s %.3f ms  nodes/ray %.1f tris/ray %.1f rays/frame %.1fM' % (d['config']['workload'].split('(')[1].split(',')[0], d['value'], d['ms_per_step'], r['nodes_per_ray'], r['tris_per_ray'], d['config']['rays_per_frame']/1e6))"
done
