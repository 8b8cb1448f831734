#!/bin/bash
# same-box A/B: the round-1 tree (git worktree _r01 = 033b5d0) against the working tree, configs 1 and 2, alternating
root=${GRAFT_REPO_ROOT:-/root/repo}
run() { ( cd $1 && shift && timeout -k 10 200 python bench.py --warmup 8 --no-cpu-baseline --no-alone "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Mrays/s %.3f ms' % (d['value'], d['ms_per_step']))" ); }
for rep in 1 2; do
  for cfg in "--config 1 --steps 64" "--config 2 --steps 32"; do
    printf "round-1 tree  %-24s" "$cfg"; run $root/_r01 $cfg
    printf "working tree  %-24s" "$cfg"; run $root $cfg
  done
done
