#!/usr/bin/env python3
"""usage: fetch_size_report.py <rocprofv3 output dir> ... : per kernel of tools/microbench/fetch_size.hip the counters' values (last
repetition), beside the byte counts known by construction (2 GiB tables)"""
import csv
import glob
import os
import sys

G = 2 << 30
TRUE = {  # kernel -> (bytes asked, bytes in whole 64-byte lines)
    "stream_read16": (G, G), "stream_write16": (G, G), "gather_rec<4>": (G, G), "gather_rec<3>": (G // 64 * 48, G),
    "gather_rec<1>": (G // 128 * 16, G // 128 * 64), "gather4": (G // 64 * 4, G),
}
vals = {}
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = {}
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            per[(r["Dispatch_Id"], k, r["Counter_Name"])] = per.get((r["Dispatch_Id"], k, r["Counter_Name"]), 0.0) + float(r["Counter_Value"])
        for (disp, k, c), v in sorted(per.items(), key=lambda kv: int(kv[0][0])):
            vals[(k, c)] = v  # the last dispatch of the kernel wins
for (k, c), v in sorted(vals.items()):
    asked, lines = TRUE.get(k, (0, 0))
    if not asked:
        continue
    print("%-16s %-10s %14.0f (counter units) = %7.3f GiB if KiB | asked %6.3f GiB, whole 64-byte lines %6.3f GiB | counter / lines = %.3f" % (
        k, c, v, v / (1 << 20), asked / (1 << 30), lines / (1 << 30), v * 1024 / lines))
