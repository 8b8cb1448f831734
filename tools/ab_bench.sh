#!/bin/bash
# usage: tools/ab_bench.sh <tag> [bench.py args...]: the bench line of the in-tree library and of rust-renderer_amd/libutopian_hip_prev.so
# (a build of an earlier commit, same box, back to back) -> gpurun_out/<tag>_new.json / <tag>_prev.json
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
for which in new prev new prev; do
  if [ $which = prev ]; then export UTOPIAN_HIP_LIB=$root/rust-renderer_amd/libutopian_hip_prev.so; else unset UTOPIAN_HIP_LIB; fi
  timeout -k 10 200 python3 bench.py --steps 64 --warmup 8 --no-cpu-baseline "$@" >> gpurun_out/${tag}_$which.json 2>> gpurun_out/${tag}_$which.err || echo "$which failed"
done
python3 - <<PY
import json
for w in ("new","prev"):
    for line in open("gpurun_out/${tag}_%s.json" % w):
        if line.startswith("{"):
            d=json.loads(line); c=d["config"]
            print(w, round(d["value"]), "Mrays/s", round(d["ms_per_step"],3), "ms | tree", round(d.get("value_tree_walk") or 0), "| pipelined", round(c["pipelined_frame_ms"] or 0,3), "interactive", round(c["interactive_frame_ms"] or 0,3), "| serial", {k: round(v,3) for k,v in (c["serial_kernel_ms_per_frame"] or {}).items()})
PY
