#!/bin/bash
# same-box comparison of several builds of the library: tools/ab_libs.sh "<bench args>" lib1.so lib2.so ...
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
args=$1; shift
for rep in 1 2; do
for lib in "$@"; do
  printf "%-26s %-44s" "$lib" "$args"; UTOPIAN_HIP_LIB=$root/rust-renderer_amd/$lib timeout -k 10 200 python bench.py --warmup 8 --no-cpu-baseline --no-alone $args 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.1f Mrays/s %.3f ms | closest %.1f shadow %.1f shade %.1f ms (HIP-event sums, overlapped)' % (d['value'], d['ms_per_step'], r.get('trace_closest_ms', 0), r.get('trace_shadow_ms', 0), r.get('shade_ms', 0)))"
done
done
