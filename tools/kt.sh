#!/bin/bash
# usage: tools/kt.sh <tag> [bench.py args...] : rocprofv3 --kernel-trace --stats of `python3 bench.py --no-cpu-baseline --no-alone <args>`;
# prints the per-kernel summary (the CSV stays under gpurun_out/<tag>_kt)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd $root
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/${tag}_kt --output-format csv -- python3 bench.py --no-cpu-baseline --no-alone "$@" > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_kt.log || { echo "kernel trace failed"; tail -5 gpurun_out/${tag}_kt.log; }
f=$(find gpurun_out/${tag}_kt -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-46s calls %4s  total %9.3f ms  avg %9.3f ms  %5.1f %%" % (r["Name"].split("(")[0].replace("void ", "").replace("uh::", "")[:46], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, float(r["Percentage"])))
PY
