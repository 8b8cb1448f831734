#!/bin/bash
# phased (one load phase per iteration) against chained steps; occupancy sensitivity of the refill kernel
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "0 8" "3 6" "3 5" "3 4" "5 6" "5 5" "5 4" "6 6" "7 6"; do
  set -- $cfg
  echo "== raw closest_variant=$1 closest_blocks_per_cu=$2"
  timeout -k 10 120 python tools/raw_trace_bench.py 5 closest_variant=$1 closest_blocks_per_cu=$2 2>&1 | grep -v amdgpu.ids
done
run() { printf "%-70s" "$*"; timeout -k 10 200 python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-alone "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.1f Mrays/s %.3f ms | closest %.3f ms/launch shadow_ms %.1f shade_ms %.1f' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['trace_shadow_ms'], r['shade_ms']))"; }
run
run --opt closest_variant=5
run --opt trace_variant=5
run --opt trace_variant=6
run --opt trace_variant=0 --opt closest_blocks_per_cu=8 --opt shadow_blocks_per_cu=8
