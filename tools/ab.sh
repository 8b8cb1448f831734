#!/bin/bash
# Same-box A/B of builds of the library: tools/ab.sh "<bench args>" lib1.so lib2.so ... ("-" = the default libutopian_hip.so), two rounds
# interleaved. Per run: the timed rate, ms per frame, the serialised per-kernel times of the bench's calibration frames (ms per frame,
# one 16-frame wavefront alone on the GPU) and the counted node / triangle visits per closest-hit ray.
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
args=$1; shift
run() {
   lib=$1
   if [ "$lib" = "-" ]; then unset UTOPIAN_HIP_LIB; else export UTOPIAN_HIP_LIB=$root/rust-renderer_amd/$lib; fi
   printf "%-34s " "$lib"
   timeout -k 10 300 python bench.py --warmup 8 --no-cpu-baseline --no-tree-walk $args 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; s = d['config'].get('serial_kernel_ms_per_frame') or {}
print('%8.1f Mrays/s %.4f ms | serial closest %.4f camera %.4f shadow %.4f shade %.4f | %.2f nodes %.2f tris per ray | interactive %.3f pipelined %.3f' % (
   d['value'], d['ms_per_step'], s.get('trace_closest', 0), s.get('camera_grid', 0), s.get('trace_shadow', 0), s.get('shade_hit_and_miss', 0), r['nodes_per_ray'], r['tris_per_ray'],
   d['config'].get('interactive_frame_ms') or 0, d['config'].get('pipelined_frame_ms') or 0))"
}
for rep in 1 2; do for lib in "$@"; do run $lib; done; done
