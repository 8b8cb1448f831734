#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/prof_*) into the summaries kept under profiles/.

  python tools/profile_summary.py <tag> [--kt DIR] [--fetch DIR] [--write DIR]

Writes profiles/<tag>_kernel_stats.csv (copy of rocprofv3 --kernel-trace --stats kernel_stats),
profiles/<tag>_pmc.json (per-kernel FETCH_SIZE / WRITE_SIZE per dispatch) and refreshes
profiles/pmc_trace_closest.json, which bench.py reads for roofline.traffic.

HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports half of the bytes of wide (16 B/lane) reads, so the read side is doubled;
WRITE_SIZE is exact for 16 B/lane stores. The two counters come from separate --pmc passes.
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    return name.split("(")[0].replace("void ", "").replace("uh::", "")


def per_kernel(dirname, counter):
    files = sorted(glob.glob(os.path.join(dirname, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]  # newest run only
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            a = agg[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return {k: {"dispatches": v[0], "sum_kib": v[1], "kib_per_dispatch": v[1] / v[0]} for k, v in agg.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--kt", default=os.path.join(ROOT, "gpurun_out", "prof_kt"))
    ap.add_argument("--fetch", default=os.path.join(ROOT, "gpurun_out", "prof_fetch"))
    ap.add_argument("--write", default=os.path.join(ROOT, "gpurun_out", "prof_write"))
    ap.add_argument("--kernel", default="k_trace_closest<false, false>")
    args = ap.parse_args()
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    ks = sorted(glob.glob(os.path.join(args.kt, "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if ks:
        shutil.copy(ks[-1], os.path.join(out, f"{args.tag}_kernel_stats.csv"))
    fetch, write = per_kernel(args.fetch, "FETCH_SIZE"), per_kernel(args.write, "WRITE_SIZE")
    pmc = {"units": "KiB per dispatch as reported by rocprofv3; hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 correction)", "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, {}), write.get(k, {})
        pmc["kernels"][k] = {
            "fetch_kib_per_dispatch": f.get("kib_per_dispatch"),
            "write_kib_per_dispatch": w.get("kib_per_dispatch"),
            "dispatches": f.get("dispatches") or w.get("dispatches"),
            "hbm_bytes_per_dispatch": 2 * 1024 * f.get("kib_per_dispatch", 0.0) + 1024 * w.get("kib_per_dispatch", 0.0),
        }
    json.dump(pmc, open(os.path.join(out, f"{args.tag}_pmc.json"), "w"), indent=1)
    if args.kernel in pmc["kernels"]:
        k = pmc["kernels"][args.kernel]
        json.dump(
            {"kernel": args.kernel, "source": f"profiles/{args.tag}_pmc.json", "hbm_bytes_per_launch": k["hbm_bytes_per_dispatch"],
             "fetch_kib_per_launch": k["fetch_kib_per_dispatch"], "write_kib_per_launch": k["write_kib_per_dispatch"]},
            open(os.path.join(out, "pmc_trace_closest.json"), "w"), indent=1,
        )
    print(open(os.path.join(out, f"{args.tag}_pmc.json")).read()[:1500])


if __name__ == "__main__":
    main()
