#!/bin/bash
# one frame per call under library options: tools/interactive_sweep3.sh "<bench args>" "name=a,name2=b" ...   ("-" = none)
cd ${GRAFT_REPO_ROOT:-/root/repo}
args=$1; shift
for rep in 1 2; do
for set in "$@"; do
  o=""; [ "$set" != "-" ] && for kv in ${set//,/ }; do o="$o --opt $kv"; done
  printf "%-44s" "$set"; timeout -k 10 300 python bench.py --no-cpu-baseline --no-tree-walk $args $o 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Mrays/s %.3f ms batched | serial %.3f | interactive %.3f | pipelined %.3f' % (d['value'], d['ms_per_step'], d['config']['frame_by_frame_ms'], d['config']['interactive_frame_ms'], d['config']['pipelined_frame_ms']))"
done
done
