#!/bin/bash
# frame time of every traversal kernel variant (0 = batch kernels, 1..4 = refill at 1 / 4 / 8 / 16 idle lanes)
for v in 0 1 2 3 4; do
  printf "closest_variant %2d  " $v
  timeout -k 10 200 python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-alone --opt closest_variant=$v 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Mrays/s %.3f ms  closest launch %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
for v in 0 1 2 3 4; do
  printf "shadow_variant %2d  " $v
  timeout -k 10 200 python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-alone --opt shadow_variant=$v 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Mrays/s %.3f ms' % (d['value'], d['ms_per_step']))"
done
