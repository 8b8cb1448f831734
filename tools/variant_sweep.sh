#!/bin/bash
# frame time of every traversal kernel variant (closest with the default shadow kernel, then shadow with the default closest kernel)
for v in 0 2 3 4 17 18 19 20 21 22 23 24 25 26; do
  printf "closest_variant %2d  " $v
  timeout -k 10 200 python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-alone --opt closest_variant=$v 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Mrays/s %.3f ms  closest launch %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
for v in 0 2 3 4 17 18 19 20 21; do
  printf "shadow_variant %2d  " $v
  timeout -k 10 200 python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-alone --opt shadow_variant=$v 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Mrays/s %.3f ms' % (d['value'], d['ms_per_step']))"
done
