#!/bin/bash
# usage: tools/opt_sweep.sh <tag> "<bench args>" opt1=val1 "opt2=val2 opt3=val3" ...: one bench line per option set (quoted sets are applied
# together; "-" = the defaults) -> gpurun_out/<tag>_sweep.txt
tag=$1; shift
base=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
: > gpurun_out/${tag}_sweep.txt
for set in "$@"; do
  opts=""
  if [ "$set" != "-" ]; then for kv in $set; do opts="$opts --opt $kv"; done; fi
  timeout -k 10 200 python3 bench.py --no-cpu-baseline $base $opts > gpurun_out/${tag}_one.json 2>> gpurun_out/${tag}_sweep.err || echo "$set failed" >> gpurun_out/${tag}_sweep.txt
  python3 - "$set" >> gpurun_out/${tag}_sweep.txt <<PY
import json, sys
for line in open("gpurun_out/${tag}_one.json"):
    if line.startswith("{"):
        d = json.loads(line); c = d["config"]
        print("%-44s %7.0f Mrays/s %6.3f ms | pipelined %.3f interactive %.3f | serial %s" % (sys.argv[1], d["value"], d["ms_per_step"], c["pipelined_frame_ms"] or 0, c["interactive_frame_ms"] or 0,
              {k: round(v, 3) for k, v in (c["serial_kernel_ms_per_frame"] or {}).items()}))
PY
done
cat gpurun_out/${tag}_sweep.txt
