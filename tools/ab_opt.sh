#!/bin/bash
# Same-box A/B of library options: tools/ab_opt.sh "<bench args>" "name=a [name2=b ...]" "name=c" ...  ("-" = defaults), two rounds interleaved.
# Per run: the timed rate, ms per frame, the serialised per-kernel times of the bench's calibration frames, one frame per call (waited for / in flight).
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
args=$1; shift
run() {
   set="$1"; opts=""
   if [ "$set" != "-" ]; then for kv in $set; do opts="$opts --opt $kv"; done; fi
   printf "%-34s " "$set"
   timeout -k 10 300 python bench.py --warmup 8 --no-cpu-baseline --no-tree-walk $args $opts 2>gpurun_out/ab_opt_last.err | tail -1 | python -c "
import sys, json
try:
   d = json.loads(sys.stdin.read())
except Exception:
   print('FAILED (gpurun_out/ab_opt_last.err)'); sys.exit(0)
s = d['config'].get('serial_kernel_ms_per_frame') or {}
print('%8.1f Mrays/s %.4f ms | serial closest %.4f camera %.4f shadow %.4f shade %.4f light %.4f | interactive %.3f pipelined %.3f' % (
   d['value'], d['ms_per_step'], s.get('trace_closest', 0), s.get('camera_grid', 0), s.get('trace_shadow', 0), s.get('shade_hit_and_miss', 0), s.get('trace_light', 0),
   d['config'].get('interactive_frame_ms') or 0, d['config'].get('pipelined_frame_ms') or 0))"
}
for rep in 1 2; do for set in "$@"; do run "$set"; done; done
