#!/bin/bash
# same-box A/B of one library option: tools/ab_opt.sh name=a name=b [bench args...]
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
A=$1; B=$2; shift 2
run() { opt=$1; shift; printf "%-28s %-40s" "$opt" "$*"; timeout -k 10 200 python bench.py --warmup 8 --no-cpu-baseline --no-tree-walk --opt $opt "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.1f Mrays/s %.3f ms | pipelined %.3f interactive %.3f' % (d['value'], d['ms_per_step'], d['config'].get('pipelined_frame_ms') or 0, d['config'].get('interactive_frame_ms') or 0))"; }
for rep in 1 2; do
for o in $A $B; do
  run $o --steps 64 "$@"
  run $o --steps 20 "$@"
done
done
for o in $A $B; do run $o --config 2 --steps 32 "$@"; done
