#!/bin/bash
# bench lines of configs 2-4 with their own bounded CPU-oracle sample
cd ${GRAFT_REPO_ROOT:-/root/repo}
tag=r02
timeout -k 10 300 python bench.py --config 2 --steps 32 --cpu-sample 1920x1080x4 > gpurun_out/${tag}_bench_config2.json 2>gpurun_out/c2.err
timeout -k 10 400 python bench.py --config 3 --width 3840 --height 2160 --steps 16 --cpu-sample 3840x2160x1 > gpurun_out/${tag}_bench_config3_4k.json 2>gpurun_out/c3.err
timeout -k 10 300 python bench.py --config 4 --steps 64 --cpu-sample 1920x1080x8 > gpurun_out/${tag}_bench_config4_isosurface.json 2>gpurun_out/c4.err
for f in gpurun_out/${tag}_bench_config[234]*.json; do tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['cpu_baseline'])"; done
