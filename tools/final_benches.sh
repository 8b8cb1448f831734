#!/bin/bash
# usage: tools/final_benches.sh <tag>: the bench lines DESIGN.md section 7 quotes -> gpurun_out/<tag>_bench_*.json (copy into profiles/)
tag=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
run() { name=$1; shift; timeout -k 10 280 python3 bench.py "$@" > gpurun_out/${tag}_bench_$name.json 2> gpurun_out/${tag}_bench_$name.err || echo "$name failed"; echo "$name done"; }
run main --steps 64 --warmup 8
run driver_style --steps 20 --warmup 4
run 8x8spp --steps 8 --warmup 2 --spp 8 --cpu-sample 1920x1080x1
run config2 --config 2 --steps 32 --warmup 8
run config0_rtiow --config 0 --width 256 --height 256 --steps 64 --warmup 8 --cpu-sample 256x256x8
run config3_4k --config 3 --width 3840 --height 2160 --steps 16 --warmup 4 --cpu-sample 3840x2160x1
run config4_isosurface --config 4 --steps 64 --warmup 4
for w in 2 4 8; do
  run emulated_world$w --steps 64 --warmup 8 --emulate-world $w --no-cpu-baseline
  run config2_emulated_world$w --config 2 --steps 32 --warmup 8 --emulate-world $w --no-cpu-baseline
done
run force_dist --steps 32 --warmup 8 --force-dist --no-cpu-baseline
run config2_force_dist --config 2 --steps 32 --warmup 8 --force-dist --no-cpu-baseline
python3 - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${tag}_bench_*.json")):
    for line in open(f):
        if line.startswith("{"):
            d = json.loads(line)
            print("%-46s %8.0f Mrays/s %8.3f ms/step  tree %s  parity %s" % (f.split("_bench_")[1][:-5], d["value"], d["ms_per_step"], round(d.get("value_tree_walk") or 0), (d.get("parity") or {}).get("max_pixel_l2")))
PY
