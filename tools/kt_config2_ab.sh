#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
for tree in ${TREES:-_r01 .}; do
  tag=$( [ "$tree" = "." ] && echo now || echo r01 )
  ( cd $root/$tree && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/kt_c2_$tag --output-format csv -- python3 bench.py --config 2 --steps 32 --warmup 4 --no-cpu-baseline --no-alone > $root/gpurun_out/kt_c2_$tag.json 2> $root/gpurun_out/kt_c2_$tag.log )
  echo "== $tag"; python3 - <<PY
import csv, glob
f = sorted(glob.glob("$root/gpurun_out/kt_c2_$tag/**/*kernel_stats.csv", recursive=True))[-1]
for r in csv.DictReader(open(f)):
    print("%-60s calls %5s total %9.3f ms avg %8.3f us" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
done
