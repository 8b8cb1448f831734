#!/bin/bash
# Same-box A/B of one frame per call: tools/ab_frame.sh "<frame_timeline.py args>" lib1.so lib2.so ... ("-" = the default library), two
# rounds interleaved; per run the interactive (a wait after every frame) and pipelined (four in flight) frame times.
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
args=$1; shift
for rep in 1 2; do for lib in "$@"; do
   if [ "$lib" = "-" ]; then unset UTOPIAN_HIP_LIB; else export UTOPIAN_HIP_LIB=$root/rust-renderer_amd/$lib; fi
   printf "%-28s " "$lib"
   timeout -k 10 120 python3 tools/frame_timeline.py --frames 24 $args 2>&1 | tail -1
done; done
