#!/bin/bash
# counters of k_trace_closest for builds of the library, on the same box: tools/pmc_variants.sh <tag prefix> lib1.so lib2.so ... ("-" = default)
# (the raw bounce rays of tools/raw_trace_bench.py: 2.07 M diffuse bounce rays of the config-1 scene, the kernel alone on the GPU)
root=${GRAFT_REPO_ROOT:-/root/repo}
prefix=$1; shift
for lib in "$@"; do
   name=$(basename $lib .so); [ "$lib" = "-" ] && name=base
   if [ "$lib" = "-" ]; then unset UTOPIAN_HIP_LIB; else export UTOPIAN_HIP_LIB=$root/rust-renderer_amd/$lib; fi
   UH_PMC_SHORT=1 $root/tools/pmc_passes.sh ${prefix}_$name python3 $root/tools/raw_trace_bench.py 5 > /dev/null
   (cd $root && python3 tools/pmc_summary.py ${prefix}_$name > /dev/null 2>&1)
done
