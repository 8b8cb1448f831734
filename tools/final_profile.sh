#!/bin/bash
# the round's committed evidence: default bench line, kernel trace + counter passes of the same command line, the other BASELINE configs
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
tag=${1:-r03}
timeout -k 10 300 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tools/profile_bench.sh $tag
# the other configs carry their own bounded CPU-oracle sample
timeout -k 10 300 python bench.py --config 2 --steps 32 --cpu-sample 1920x1080x4 > gpurun_out/${tag}_bench_config2.json 2>/dev/null
timeout -k 10 400 python bench.py --config 3 --width 3840 --height 2160 --steps 16 --cpu-sample 3840x2160x1 > gpurun_out/${tag}_bench_config3_4k.json 2>/dev/null
timeout -k 10 300 python bench.py --config 4 --steps 64 --cpu-sample 1920x1080x8 > gpurun_out/${tag}_bench_config4_isosurface.json 2>/dev/null
timeout -k 10 300 python bench.py --config 0 --width 256 --height 256 --steps 64 --cpu-sample 256x256x64 > gpurun_out/${tag}_bench_config0_rtiow.json 2>/dev/null
for w in 2 4 8; do timeout -k 10 200 python bench.py --emulate-world $w --no-cpu-baseline > gpurun_out/${tag}_bench_emulated_world$w.json 2>/dev/null; done
timeout -k 10 200 python bench.py --force-dist --no-cpu-baseline --steps 16 > gpurun_out/${tag}_bench_force_dist.json 2> gpurun_out/${tag}_bench_force_dist.err
# SURVEY 8d's second reporting forms; config 2 on N ranks (reservoir passes by bands of rows against full-frame passes on every rank); RCCL link with one rank
timeout -k 10 300 python bench.py --spp 8 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_bench_8x8spp.json 2>/dev/null
timeout -k 10 300 python bench.py --config 2 --steps 32 --no-cpu-baseline --opt full_frame_restir=1 > gpurun_out/${tag}_bench_config2_full_frame_restir.json 2>/dev/null
for w in 2 4 8; do
  timeout -k 10 300 python bench.py --config 2 --steps 32 --emulate-world $w --no-cpu-baseline > gpurun_out/${tag}_bench_config2_emulated_world$w.json 2>/dev/null
  timeout -k 10 300 python bench.py --config 2 --steps 32 --emulate-world $w --no-cpu-baseline --full-frame-reservoir-passes > gpurun_out/${tag}_bench_config2_emulated_world${w}_full_frame_passes.json 2>/dev/null
done
timeout -k 10 300 python bench.py --config 2 --steps 32 --force-dist --no-cpu-baseline > gpurun_out/${tag}_bench_config2_force_dist.json 2> gpurun_out/${tag}_bench_config2_force_dist.err
for f in gpurun_out/${tag}_bench*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-52s %9.1f %s  %8.3f ms/step  frame-by-frame %s" % (sys.argv[1].split("/")[-1], d["value"], d["unit"], d["ms_per_step"], d["config"].get("frame_by_frame_ms")))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
