#!/bin/bash
# usage: tools/timeline.sh <tag> [frame_timeline.py args...] : the kernel trace of one-frame-per-call frames, then the last frame's timeline
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd $root
rm -rf gpurun_out/${tag}_tl
timeout -k 10 300 rocprofv3 --kernel-trace -d $root/gpurun_out/${tag}_tl --output-format csv -- python3 tools/frame_timeline.py "$@" > gpurun_out/${tag}_tl.log 2>&1 || { echo "trace failed"; tail -5 gpurun_out/${tag}_tl.log; }
grep interactive gpurun_out/${tag}_tl.log
python3 tools/frame_timeline.py --report gpurun_out/${tag}_tl
