#!/bin/bash
# usage: tools/profile_bench.sh <tag> [bench.py args...]
# (1) rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-alone <args>`
#     (its bench line goes to gpurun_out/<tag>_bench_under_rocprof.json: HIP-event launch time beside the trace's);
# (2) the counter passes of tools/pmc_passes.sh on `python3 bench.py --steps 48 --warmup 16 --no-cpu-baseline --no-alone <args>`.
# Afterwards, off the GPU box: python tools/pmc_summary.py <tag> --bench ; cp gpurun_out/<tag>_kt/**/*kernel_stats.csv profiles/
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
export UH_BENCH_SIGNATURE=$root/gpurun_out/bench_signature.json
cd $root
rm -rf $root/gpurun_out/${tag}_kt
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/${tag}_kt --output-format csv -- python3 bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-alone --no-tree-walk "$@" > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_kt.log || { echo "kernel trace failed"; tail -5 gpurun_out/${tag}_kt.log; }
tools/pmc_passes.sh $tag python3 bench.py --steps 48 --warmup 16 --no-cpu-baseline --no-alone --no-tree-walk "$@"
