#!/bin/bash
# same-box A/B of library options (and builds) with the serialised per-kernel times:
#   tools/ab_opt_kernels.sh [lib.so:]name=value ... [-- bench args]      (lib.so: a file under rust-renderer_amd/, default the current build)
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
opts=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do opts+=("$1"); shift; done
[ "$1" == "--" ] && shift
for rep in 1 2; do
for spec in "${opts[@]}"; do
  lib=libutopian_hip.so; o=$spec
  case $spec in *:*) lib=${spec%%:*}; o=${spec#*:};; esac
  printf "%-44s" "$spec"
  UTOPIAN_HIP_LIB=$root/rust-renderer_amd/$lib timeout -k 10 200 python bench.py --warmup 8 --steps 64 --no-cpu-baseline --no-tree-walk --opt $o "$@" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['config']['serial_kernel_ms_per_frame']
print('%.1f Mrays/s %.3f ms | ' % (d['value'], d['ms_per_step']) + ' '.join('%s %.4f' % (a, b) for a, b in k.items()) + ' | pipelined %.3f interactive %.3f | sun grid build %.2f ms, %d entries; with builds %.0f' % (d['config']['pipelined_frame_ms'], d['config']['interactive_frame_ms'], d['sun_grid']['build_ms'], d['sun_grid']['entries'], d['value_with_grid_builds']))"
done
done
