#!/bin/bash
# LDS part of the traversal stack (16 / 12 / 10 / 8 entries per lane) against the blocks per CU it admits
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
run() { lib=$1; shift; printf "%-24s %-78s" "$lib" "$*"; UTOPIAN_HIP_LIB=$root/rust-renderer_amd/$lib timeout -k 10 200 python bench.py --warmup 8 --no-cpu-baseline --no-alone "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.1f Mrays/s %.3f ms | closest %.3f ms/launch' % (d['value'], d['ms_per_step'], r['avg_launch_ms']))"; }
for rep in 1 2; do
run libutopian_hip.so --steps 64
run libutopian_hip_st12.so --steps 64
run libutopian_hip_st12.so --steps 64 --opt closest_blocks_per_cu=7 --opt shadow_blocks_per_cu=6
run libutopian_hip_st10.so --steps 64 --opt closest_blocks_per_cu=8 --opt shadow_blocks_per_cu=6
run libutopian_hip_st8.so --steps 64 --opt closest_blocks_per_cu=8 --opt shadow_blocks_per_cu=7
run libutopian_hip_st8.so --steps 64
done
run libutopian_hip.so --config 2 --steps 32
run libutopian_hip_st12.so --config 2 --steps 32 --opt closest_blocks_per_cu=7 --opt shadow_blocks_per_cu=6
run libutopian_hip_st10.so --config 2 --steps 32 --opt closest_blocks_per_cu=8 --opt shadow_blocks_per_cu=6
run libutopian_hip_st8.so --config 2 --steps 32 --opt closest_blocks_per_cu=8 --opt shadow_blocks_per_cu=7
