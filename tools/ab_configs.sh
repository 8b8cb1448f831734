#!/bin/bash
# same-box comparison of two builds of the library over the BASELINE configs: tools/ab_configs.sh [libA.so libB.so]
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
A=${1:-libutopian_hip_prev.so}; B=${2:-libutopian_hip.so}
run() { lib=$1; shift; printf "%-26s %-58s" "$lib" "$*"; UTOPIAN_HIP_LIB=$root/rust-renderer_amd/$lib timeout -k 10 300 python bench.py --warmup 8 --no-cpu-baseline --no-alone --no-tree-walk "$@" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); g=d.get('sun_grid') or {}
print('%.1f Mrays/s %.3f ms | sun grid %s build %.2f ms' % (d['value'], d['ms_per_step'], 'in use' if g.get('in_use') else 'not in use', g.get('build_ms') or 0))"; }
for lib in $A $B; do run $lib --steps 20 --warmup 4; done
for lib in $A $B; do run $lib --config 2 --steps 32; done
for lib in $A $B; do run $lib --config 3 --width 3840 --height 2160 --steps 16; done
for lib in $A $B; do run $lib --config 4 --steps 64; done
for lib in $A $B; do run $lib --config 0 --width 256 --height 256 --steps 64; done
for lib in $A $B; do run $lib --emulate-world 8 --steps 64; done
for lib in $A $B; do run $lib --config 2 --emulate-world 8 --steps 32; done
