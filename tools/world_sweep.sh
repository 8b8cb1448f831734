#!/bin/bash
# one rank's share of an N-rank tile partition on one GPU, defaults (and the frames-per-wavefront / in-flight alternatives)
run() { printf "%-60s" "$*"; timeout -k 10 200 python bench.py --steps 64 --warmup 4 --no-cpu-baseline --no-alone "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Mrays/s %.3f ms' % (d['value'], d['ms_per_step']))"; }
run
run --emulate-world 2
run --emulate-world 4
run --emulate-world 8
run --emulate-world 8 --opt frames_in_flight=3
run --emulate-world 8 --opt batch_frames=2
run --emulate-world 8 --opt batch_frames=8
run --config 2 --steps 16
run --config 2 --steps 16 --opt frames_in_flight=3
