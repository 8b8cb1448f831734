#!/usr/bin/env python3
"""Condense the rocprofv3 --pmc passes of tools/pmc_passes.sh into profiles/<tag>_counters.json:
per kernel, per counter: median and mean over the kernel's dispatches, plus derived figures."""
import collections
import csv
import glob
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    return name.split("(")[0].replace("void ", "").replace("uh::", "")


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}")
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, "**", "*_counter_collection.csv"), recursive=True):
        per_dispatch = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            per_dispatch[(r["Dispatch_Id"], short(r["Kernel_Name"]), r["Counter_Name"])] += float(r["Counter_Value"])
        for (_, k, c), v in per_dispatch.items():
            vals[k][c].append(v)
    out = {"note": "rocprofv3 --pmc, one process per pass (tools/pmc_passes.sh); per kernel and counter: median / mean over its dispatches. "
                   "lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU); valu_issue_frac = SQ_INSTS_VALU * 2 clk / (1024 SIMDs * GRBM_GUI_ACTIVE/8); "
                   "vmem_latency_clk = 4 * SQ_INST_LEVEL_VMEM / SQ_INSTS_VMEM_RD (level counters tick in quad-cycles)", "kernels": {}}
    for k, cs in sorted(vals.items()):
        e = {c: {"median": statistics.median(v), "mean": sum(v) / len(v), "n": len(v)} for c, v in sorted(cs.items())}
        m = {c: e[c]["median"] for c in e}
        d = {}
        if m.get("SQ_ACTIVE_INST_VALU"):
            d["lane_utilisation"] = m.get("SQ_THREAD_CYCLES_VALU", 0) / (64.0 * m["SQ_ACTIVE_INST_VALU"])
        if m.get("GRBM_GUI_ACTIVE") and m.get("SQ_INSTS_VALU"):
            d["valu_issue_frac"] = m["SQ_INSTS_VALU"] * 2.0 / (1024.0 * m["GRBM_GUI_ACTIVE"] / 8.0)
        if m.get("SQ_WAVE_CYCLES"):
            d["wait_any_share"] = m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"]
            d["wait_inst_share"] = m.get("SQ_WAIT_INST_ANY", 0) / m["SQ_WAVE_CYCLES"]
        if m.get("SQ_INSTS_VMEM_RD") and m.get("SQ_INST_LEVEL_VMEM"):
            d["vmem_latency_clk"] = 4.0 * m["SQ_INST_LEVEL_VMEM"] / m["SQ_INSTS_VMEM_RD"]
        if m.get("TCC_HIT_sum") is not None and (m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0)) > 0:
            d["l2_hit"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
        if m.get("TCP_TCC_READ_REQ_sum") and m.get("TCP_TCC_READ_REQ_LATENCY_sum"):
            d["l1_to_l2_read_latency_clk"] = m["TCP_TCC_READ_REQ_LATENCY_sum"] / m["TCP_TCC_READ_REQ_sum"]
        if m.get("TCP_TOTAL_CACHE_ACCESSES_sum") and m.get("TCP_TCC_READ_REQ_sum"):
            d["l1_miss_per_access"] = m["TCP_TCC_READ_REQ_sum"] / m["TCP_TOTAL_CACHE_ACCESSES_sum"]
        out["kernels"][k] = {"counters": e, "derived": d}
    dst = os.path.join(ROOT, "profiles", f"{tag}_counters.json")
    json.dump(out, open(dst, "w"), indent=1)
    for k, v in out["kernels"].items():
        print(k, json.dumps({a: round(b, 4) for a, b in v["derived"].items()}))
        print("   ", {c: round(x["median"]) for c, x in v["counters"].items()})


if __name__ == "__main__":
    main()
