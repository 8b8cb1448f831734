#!/usr/bin/env python3
"""One frame per call with a wait after every frame (a moving camera: main.rs:460-471) - default options, side stream on: under
`rocprofv3 --kernel-trace` the last frame's dispatches show where the frame's time goes (start offset, duration, what overlaps).
  rocprofv3 --kernel-trace -d gpurun_out/tl_kt --output-format csv -- python3 tools/frame_timeline.py [--opt name=value ...]
  python3 tools/frame_timeline.py --report gpurun_out/tl_kt"""
import argparse
import csv
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def report(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    start = max(i for i, r in enumerate(rows) if "k_generate" in r["Kernel_Name"])
    t0 = int(rows[start]["Start_Timestamp"])
    busy_until = t0
    idle = 0
    for r in rows[start:]:
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if a > busy_until:
            idle += a - busy_until
        busy_until = max(busy_until, b)
        print("%-30s start %8.3f  dur %8.3f ms  grid %s" % (r["Kernel_Name"].split("(")[0].replace("void ", "").replace("uh::", "")[:30], (a - t0) / 1e6, (b - a) / 1e6, r.get("Grid_Size", "")))
    print("first start to last end %.3f ms; nothing running for %.3f ms of it" % ((busy_until - t0) / 1e6, idle / 1e6))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=1)
    ap.add_argument("--tex-size", type=int, default=1024)
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--report")
    a = ap.parse_args()
    if a.report:
        return report(a.report)
    import rust_renderer_amd as rr
    W, H = a.width, a.height
    scene = rr.scenes.scene_for_config(a.config, tex_size=a.tex_size)
    r = rr.Renderer(W, H)
    for kv in a.opt:
        k, v = kv.split("=")
        r.set_option(k, int(v))
    scene.upload(r)
    mask = rr.PASS_ALL if a.config == 2 else rr.PASS_REFERENCE_PT
    view_kw = {}  # (the scene's own view flags: scenes.scene_for_config)
    loop = rr.FrameLoop(r, scene.make_view(W, H, **view_kw))
    for _ in range(4):
        loop.frame(mask)
        r.synchronize()
    t = time.perf_counter()
    for _ in range(4 * a.frames):  # without a wait: four frames in flight
        loop.frame(mask)
    r.synchronize()
    pipelined = (time.perf_counter() - t) / (4 * a.frames) * 1e3
    t = time.perf_counter()
    for _ in range(a.frames):  # (last: the report reads the trace's last frame)
        loop.frame(mask)
        r.synchronize()
    print("interactive %.3f ms per frame, pipelined %.3f" % ((time.perf_counter() - t) / a.frames * 1e3, pipelined), file=sys.stderr)


if __name__ == "__main__":
    main()
