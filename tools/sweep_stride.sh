#!/bin/bash
# record stride A/B: 48-byte nodes / triangle packets at a 48-byte stride (default) against a 64-byte stride (one cache sector per record)
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
run() { lib=$1; shift; printf "%-22s %-36s" "$lib" "$*"; UTOPIAN_HIP_LIB=$root/rust-renderer_amd/$lib timeout -k 10 200 python bench.py --warmup 8 --no-cpu-baseline --no-alone "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.1f Mrays/s %.3f ms | closest %.3f ms/launch' % (d['value'], d['ms_per_step'], r['avg_launch_ms']))"; }
for rep in 1 2; do
for lib in libutopian_hip.so libutopian_hip_s44.so libutopian_hip_s43.so libutopian_hip_s34.so; do
  run $lib --steps 64
done
done
for lib in libutopian_hip.so libutopian_hip_s44.so; do run $lib --config 2 --steps 32; run $lib --config 3 --width 3840 --height 2160 --steps 16; done
UTOPIAN_HIP_LIB=$root/rust-renderer_amd/libutopian_hip_s44.so timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "matches_oracle or device_build or refit" 2>&1 | tail -2
