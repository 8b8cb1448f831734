#!/usr/bin/env python3
"""One 16-frame wavefront of config 1 (or --config N) with nothing overlapped (one stream, one frame in flight): run under
`rocprofv3 --kernel-trace` its dispatches show what every launch of the chain costs alone on the GPU, bounce by bounce.
  rocprofv3 --kernel-trace -d gpurun_out/serial_kt --output-format csv -- python3 tools/serial_wavefront.py [--opt name=value ...]
  python3 tools/serial_wavefront.py --report gpurun_out/serial_kt"""
import argparse
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def report(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # the last wavefront: everything from the last k_generate on
    start = max(i for i, r in enumerate(rows) if "k_generate" in r["Kernel_Name"])
    total = 0.0
    for r in rows[start:]:
        ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        total += ns
        print("%-34s %9.3f ms   grid %s" % (r["Kernel_Name"].split("(")[0].replace("void ", "").replace("uh::", "")[:34], ns / 1e6, r.get("Grid_Size", "")))
    span = int(rows[-1]["End_Timestamp"]) - int(rows[start]["Start_Timestamp"])
    print("sum %.3f ms per wavefront; first start to last end %.3f ms (the difference is what the GPU idles between launches)" % (total / 1e6, span / 1e6))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=1)
    ap.add_argument("--tex-size", type=int, default=1024)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--report")
    a = ap.parse_args()
    if a.report:
        return report(a.report)
    import rust_renderer_amd as rr

    W, H = 1920, 1080
    scene = rr.scenes.scene_for_config(a.config, tex_size=a.tex_size)
    r = rr.Renderer(W, H)
    for kv in a.opt:
        k, v = kv.split("=")
        r.set_option(k, int(v))
    scene.upload(r)
    for k, v in (("frames_in_flight", 1), ("overlap", 0)):
        r.set_option(k, v)
    mask = rr.PASS_ALL if a.config == 2 else rr.PASS_REFERENCE_PT
    loop = rr.FrameLoop(r, scene.make_view(W, H))
    loop.frames(a.frames, mask)
    r.synchronize()
    loop.frames(a.frames, mask)
    r.synchronize()
    s = r.get_stats()
    print("rays per frame:", [x // (2 * a.frames) for x in s.rays], file=sys.stderr)


if __name__ == "__main__":
    main()
