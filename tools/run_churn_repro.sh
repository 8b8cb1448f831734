#!/bin/bash
# The library-free reproducer (tools/hip_churn_repro.cpp) under both HIP runtimes, four processes side by side each (how round 4's
# soaks met the corruption soonest): usage  tools/run_churn_repro.sh [seconds per leg, default 90]
# Legs: device code by hipcc 7.2 / no device code at all (g++), each against /opt/rocm's runtime and against the PyTorch wheel's copy.
# Writes gpurun_out/churn/<leg>_<n>.log and a summary line per leg; never more than four GPU processes at once.
set -u
T=${1:-90}
OUT=gpurun_out/churn
mkdir -p $OUT
WHEEL=$(python3 -c "import importlib.util,os; s=importlib.util.find_spec('torch'); print(os.path.join(os.path.dirname(s.origin),'lib','libamdhip64.so') if s else '')")
hipcc -O2 --offload-arch=gfx950 -DWITH_KERNEL tools/hip_churn_repro.cpp -o $OUT/repro_kernel || exit 1
g++ -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tools/hip_churn_repro.cpp -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -lamdhip64 -o $OUT/repro_plain || exit 1
leg() {  # name binary preload
   local name=$1 bin=$2 pre=$3 pids=() rc=() n
   for n in 1 2 3 4; do
      ( export MALLOC_CHECK_=3; [ -n "$pre" ] && export LD_PRELOAD=$pre; timeout -k 10 $((T + 60)) $bin $T $n > $OUT/${name}_$n.log 2>&1 ) &
      pids+=($!)
   done
   for n in 0 1 2 3; do wait ${pids[$n]}; rc+=($?); done
   echo "$name: exit codes ${rc[*]} | $(grep -h -e '^clean' -e 'CHECK FAILED' -e 'HIP_VERSION' $OUT/${name}_1.log | tr '\n' ';')" | tee -a $OUT/summary.txt
}
: > $OUT/summary.txt
leg system_kernel $OUT/repro_kernel ""
[ -n "$WHEEL" ] && [ -e "$WHEEL" ] && leg wheel_kernel $OUT/repro_kernel "$WHEEL"
[ -n "$WHEEL" ] && [ -e "$WHEEL" ] && leg wheel_plain $OUT/repro_plain "$WHEEL"
leg system_plain $OUT/repro_plain ""
cat $OUT/summary.txt
