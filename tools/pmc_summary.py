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


def base_name(k):
    """kernel name without its template arguments: the variants of one kernel share a profile entry - except the shadow walk of the LIGHT rays
    (k_trace_shadow<COUNT, true>), which is a kernel of its own weight in config 2 (bench.py's roofline names it k_trace_shadow_light)"""
    b = k.split("<")[0].strip()
    if b == "k_trace_shadow" and "<" in k and k.split("<")[1].split(">")[0].split(",")[-1].strip() in ("true", "1"):
        return "k_trace_shadow_light"
    return b


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}")
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    durations = collections.defaultdict(lambda: collections.defaultdict(list))  # kernel -> pass file -> dispatch durations (ns)
    # one file per pass directory - the newest: gpurun merges every call's output into gpurun_out/, so a tag that was profiled
    # twice holds both runs' files side by side, and mixing them double-counts every dispatch
    newest = {}
    for f in glob.glob(os.path.join(src, "**", "*_counter_collection.csv"), recursive=True):
        d = os.path.relpath(f, src).split(os.sep)[0]
        if d not in newest or os.path.getmtime(f) > os.path.getmtime(newest[d]):
            newest[d] = f
    for f in sorted(newest.values()):
        per_dispatch = collections.defaultdict(float)
        span = {}
        for r in csv.DictReader(open(f)):
            per_dispatch[(r["Dispatch_Id"], short(r["Kernel_Name"]), r["Counter_Name"])] += float(r["Counter_Value"])
            if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                span[(r["Dispatch_Id"], short(r["Kernel_Name"]))] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        for (_, k, c), v in per_dispatch.items():
            vals[k][c].append(v)
        for (_, k), ns in span.items():
            durations[k][f].append(ns)
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
        # dispatch duration under the counter passes (kernels serialised by the profiler): the counter CSVs carry start / end
        # timestamps per dispatch, so no clock has to be assumed. Median over the dispatches of the pass with the most of them.
        ns = max(durations[k].values(), key=len) if durations.get(k) else []
        if ns:
            d["launch_ns_median"], d["launch_ns_mean"], d["launch_ns_total"], d["launches"] = statistics.median(ns), sum(ns) / len(ns), sum(ns), len(ns)
        out["kernels"][k] = {"counters": e, "derived": d}
    dst = os.path.join(ROOT, "profiles", f"{tag}_counters.json")
    json.dump(out, open(dst, "w"), indent=1)
    if "--bench" in sys.argv:
        # profiles/bench_counters.json: what bench.py quotes (roofline.traffic / roofline.valu) when the signature of its
        # run equals the one of the profiled command line (gpurun_out/bench_signature.json, written by that run)
        sig = json.load(open(os.path.join(ROOT, "gpurun_out", "bench_signature.json")))
        bench = {"signature": sig, "source": f"profiles/{tag}_counters.json (rocprofv3 --pmc passes, tools/pmc_passes.sh)",
                 "note": "per launch, median over the kernel's dispatches; hbm_bytes = 2*FETCH_SIZE KiB + WRITE_SIZE KiB (MI355X_MICROARCH.md HBM: FETCH_SIZE reports half the bytes "
                         "of wide reads on gfx950 - calibrated for coalesced 16-B loads; for the 64-B node gathers of the traversal kernels the x2 may overstate by up to 2x, "
                         "so hbm_bytes is an upper bound there); Infinity-Cache hits are counted in", "kernels": {}}
        # rays per k_trace_closest launch of the profiled run (the bench line its first pass printed): bench.py scales the
        # per-launch traffic by its own rays per launch
        profiled_rays = profiled_rays_per_frame = None
        profiled_kernel = "k_trace_closest"
        try:
            for line in open(os.path.join(src, "pass1.log")):
                if line.startswith("{") and '"roofline"' in line:
                    bl = json.loads(line)
                    profiled_rays = bl["roofline"]["rays_per_launch"]
                    profiled_kernel = bl["roofline"].get("kernel", "k_trace_closest")
                    profiled_rays_per_frame = bl.get("config", {}).get("rays_per_frame")
        except OSError:
            pass
        # HBM-side bytes of the WHOLE profiled run, every kernel and every dispatch (sum, not median x count), per frame the run
        # rendered: bench.py's roofline.frame_hbm_frac = this / ms_per_step / 8 TB/s
        total_bytes = uncorrected = 0.0
        for k, v in out["kernels"].items():
            c = v["counters"]
            fetch = 1024.0 * c.get("FETCH_SIZE", {}).get("mean", 0.0) * c.get("FETCH_SIZE", {}).get("n", 0)
            write = 1024.0 * c.get("WRITE_SIZE", {}).get("mean", 0.0) * c.get("WRITE_SIZE", {}).get("n", 0)
            total_bytes += 2.0 * fetch + write
            uncorrected += fetch + write
        bench["run_hbm_bytes_total"] = total_bytes
        if sig.get("frames_total"):
            bench["frame_hbm_bytes"] = total_bytes / sig["frames_total"]
            bench["frame_hbm_bytes_uncorrected"] = uncorrected / sig["frames_total"]  # FETCH_SIZE as reported (64 B per TCC_EA0_RDREQ): the lower bound
            bench["frames_profiled"] = sig["frames_total"]
        if profiled_rays_per_frame:
            bench["rays_per_frame"] = profiled_rays_per_frame  # path rays per frame of the profiled run: bench.py scales frame_hbm_bytes by its own
        bench["signature"] = {a: b for a, b in sig.items() if a != "frames_total"}
        for k, v in out["kernels"].items():
            m = {c: x["median"] for c, x in v["counters"].items()}
            e = bench["kernels"].setdefault(base_name(k), {})
            if e:
                continue  # first variant seen wins (one variant per run)
            if v["derived"].get("launch_ns_median"):
                e["launch_ns"] = v["derived"]["launch_ns_median"]  # serialised (counter passes run one kernel at a time)
            if "FETCH_SIZE" in m or "WRITE_SIZE" in m:
                e["fetch_kib_per_launch"], e["write_kib_per_launch"] = m.get("FETCH_SIZE"), m.get("WRITE_SIZE")
                e["hbm_bytes_per_launch"] = 2048.0 * m.get("FETCH_SIZE", 0.0) + 1024.0 * m.get("WRITE_SIZE", 0.0)
                e["hbm_bytes_per_launch_uncorrected"] = 1024.0 * m.get("FETCH_SIZE", 0.0) + 1024.0 * m.get("WRITE_SIZE", 0.0)
            if m.get("SQ_INSTS_VALU"):
                e["wave_instr_per_launch"] = m["SQ_INSTS_VALU"]
            d = v["derived"]
            e["issue_frac"], e["lane_utilisation"] = d.get("valu_issue_frac"), d.get("lane_utilisation")
            if m.get("GRBM_GUI_ACTIVE"):
                clk = m["GRBM_GUI_ACTIVE"] / 8.0
                e["ta_busy_frac"] = m.get("TA_TA_BUSY_sum", 0.0) / 256.0 / clk if m.get("TA_TA_BUSY_sum") else None
                e["td_busy_frac"] = m.get("TD_TD_BUSY_sum", 0.0) / 256.0 / clk if m.get("TD_TD_BUSY_sum") else None
                e["launch_clk"] = clk
            e["variant"] = k
            if base_name(k) == profiled_kernel and profiled_rays:
                e["rays_per_launch"] = profiled_rays  # of the kernel the profiled run's roofline names
        # one file, one profile per command line (signature): this one replaces its predecessor of the same signature
        path = os.path.join(ROOT, "profiles", "bench_counters.json")
        try:
            old = json.load(open(path))
            profiles = old.get("profiles", [old] if old.get("signature") else [])
        except Exception:
            profiles = []
        profiles = [p for p in profiles if p.get("signature") != bench["signature"]] + [bench]
        json.dump({"note": "counter profiles of bench.py command lines, keyed by signature (bench.py load_profile)", "profiles": profiles}, open(path, "w"), indent=1)
    for k, v in out["kernels"].items():
        print(k, json.dumps({a: round(b, 4) for a, b in v["derived"].items()}))
        print("   ", {c: round(x["median"]) for c, x in v["counters"].items()})


if __name__ == "__main__":
    main()
