#!/bin/bash
# same-box runs of bench.py with sets of library options: tools/ab_opt_any.sh "<bench args>" "name=a,name2=b" "name=c" ...   ("-" = no option)
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
args=$1; shift
for rep in 1 2; do
for set in "$@"; do
  o=""; [ "$set" != "-" ] && for kv in ${set//,/ }; do o="$o --opt $kv"; done
  printf "%-58s" "$set"
  timeout -k 10 300 python bench.py --warmup 8 --no-cpu-baseline --no-alone --no-tree-walk $args $o 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); g=d.get('sun_grid') or {}
print('%.1f Mrays/s %.3f ms | sun grid %s build %.2f ms mean list %.2f entries %s tests/ray %.2f to tree %.3f' % (d['value'], d['ms_per_step'], 'in use' if g.get('in_use') else 'NOT in use', g.get('build_ms') or 0, g.get('mean_list') or 0, g.get('entries'), g.get('tests_per_ray') or 0, g.get('handed_to_tree') or 0))"
done
done
