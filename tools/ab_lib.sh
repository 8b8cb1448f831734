#!/bin/bash
# same-box A/B of two builds of the library: libutopian_hip_prev.so (a copy of the previous build) against libutopian_hip.so
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
run() { lib=$1; shift; printf "%-26s %-50s" "$lib" "$*"; UTOPIAN_HIP_LIB=$root/rust-renderer_amd/$lib timeout -k 10 200 python bench.py --warmup 8 --no-cpu-baseline --no-alone "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.1f Mrays/s %.3f ms | closest %.3f ms/launch' % (d['value'], d['ms_per_step'], r['avg_launch_ms']))"; }
for rep in 1 2; do
for lib in libutopian_hip_prev.so libutopian_hip.so; do
  run $lib --steps 64
  run $lib --config 2 --steps 32
done
done
for lib in libutopian_hip_prev.so libutopian_hip.so; do run $lib --config 3 --width 3840 --height 2160 --steps 16; run $lib --emulate-world 8 --steps 64; done
