#!/bin/bash
# usage: tools/pmc_passes.sh <tag> <program and args...>   (the program itself, e.g. python3 tools/raw_trace_bench.py 5)
# Runs the rocprofv3 counter passes (one process per pass; --pmc never combined with a trace) and writes
# gpurun_out/pmc_<tag>/pass<k>; tools/pmc_summary.py condenses them into one JSON per tag.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
[ -z "$GRAFT_REPO_ROOT" ] && out=/root/repo/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out  # a tag profiled twice must not keep the earlier run's files
k=0
# UH_PMC_SHORT=1: three passes (instruction counts and lane utilisation, texture addresser / data, L2 hits) - for A/B runs of kernel variants
if [ -n "$UH_PMC_SHORT" ]; then LIST=$'SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU\nTA_TA_BUSY_sum TD_TD_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_FLAT_READ_WAVEFRONTS_sum\nTCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum'; else LIST=$(cat <<'FULL'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE
SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAVES SQ_INSTS_VALU_TRANS_F32
TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum SQ_INST_LEVEL_LDS
FETCH_SIZE
WRITE_SIZE
FULL
); fi
while IFS= read -r counters; do
  [ -z "$counters" ] && continue
  k=$((k+1))
  (cd ${GRAFT_REPO_ROOT:-/root/repo} && timeout -k 10 300 rocprofv3 --pmc $counters -d $out/pass$k --output-format csv -- "$@" > $out/pass$k.log 2>&1) || { echo "pass $k failed: $counters"; tail -5 $out/pass$k.log; }
done <<< "$LIST"
echo "pmc passes done: $out"
