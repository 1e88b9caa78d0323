#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (written by tools/profile_round.sh on the GPU box) into the tracked evidence:

  profiles/r01_bench_<tag>.json          the bench line of that run
  profiles/r01_kernel_stats_<tag>.csv    rocprofv3 --kernel-trace --stats summary (kmx kernels + copies)
  profiles/r01_pmc_summary_<tag>.json    per-kernel FETCH_SIZE / WRITE_SIZE (separate passes), corrected as
                                         MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE x2), per launch
  profiles/pmc_summary_current.json      what bench.py reads `roofline.traffic` from

Usage: python tools/summarise_profiles.py <tag> [--round r01]
"""
import argparse
import csv
import json
import os
import re
import shutil
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def counters(path, counter):
    per = defaultdict(list)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter and "kmx::" in row["Kernel_Name"]:
                per[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--round", default="r04")
    ap.add_argument("--config", type=int, default=2)
    a = ap.parse_args()
    sfx = "" if a.config == 2 else f"_cfg{a.config}"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{a.tag}{sfx}")
    a.tag += sfx
    dst = os.path.join(ROOT, "profiles")
    bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
    with open(os.path.join(dst, f"{a.round}_bench_{a.tag}.json"), "w") as f:
        json.dump(bench, f, indent=1)
    shutil.copy(os.path.join(src, "kt", "kt_kernel_stats.csv"), os.path.join(dst, f"{a.round}_kernel_stats_{a.tag}.csv"))

    avg_ns = {}
    with open(os.path.join(src, "kt", "kt_kernel_stats.csv"), newline="") as f:
        for row in csv.DictReader(f):
            avg_ns[short(row["Name"])] = (float(row["AverageNs"]), int(row["Calls"]))

    fetch = counters(os.path.join(src, "fetch", "fetch_counter_collection.csv"), "FETCH_SIZE")
    write = counters(os.path.join(src, "write", "write_counter_collection.csv"), "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        fk = sum(fetch.get(k, [0])) / max(1, len(fetch.get(k, [])))
        wk = sum(write.get(k, [0])) / max(1, len(write.get(k, [])))
        kernels[k] = {
            "launches": len(fetch.get(k, [])),
            "FETCH_SIZE_KiB_mean": round(fk, 1),
            "WRITE_SIZE_KiB_mean": round(wk, 1),
            "fetch_bytes_corrected": int(fk * 1024 * 2),
            "fetch_bytes_raw": int(fk * 1024),
            "write_bytes": int(wk * 1024),
            "hbm_bytes_per_launch": int(fk * 1024 * 2 + wk * 1024),
            "avg_launch_ms_kernel_trace": round(avg_ns[k][0] / 1e6, 4) if k in avg_ns else None,
        }
    # calibration of the gfx950 FETCH_SIZE correction in the same run: the largest k_scan_reduce launch reads exactly
    # 4 bytes per item of a grid that is a whole number of 4096-item tiles (the histogram scan of the index build:
    # 4^10 keys; before the search path lost its reduce launch, the 1e7-query scan)
    red, red_items = 0.0, 0
    with open(os.path.join(src, "fetch", "fetch_counter_collection.csv"), newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == "FETCH_SIZE" and "k_scan_reduce" in row["Kernel_Name"] and float(row["Counter_Value"]) > red:
                red, red_items = float(row["Counter_Value"]), int(row["Grid_Size"]) // 256 * 4096
    fill = max((k for k in kernels if k.startswith("kmx::k_fill<")), key=lambda k: kernels[k]["hbm_bytes_per_launch"] * kernels[k]["launches"])
    out = {
        "source": f"tools/profile_round.sh {a.tag}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) and "
                  "--kernel-trace --stats, each over `python3 bench.py" + (f" --config {a.config}" if a.config != 2 else "") + " --no-cpu-baseline --no-open-compare --no-two-streams --no-other-configs --no-host-api --steps 8 --warmup 2`, MI355X",
        "units": "FETCH_SIZE / WRITE_SIZE are KiB as reported; fetch_bytes_corrected doubles FETCH_SIZE (gfx950 tallies 128-B "
                 "requests at 64 B, MI355X_MICROARCH.md HBM section); calibration in the same run: the largest k_scan_reduce "
                 f"launch reads exactly 4 B x {red_items} items = {4 * red_items} B",
        "bench_value_M_queries_per_s": bench["value"],
        "kernels": kernels,
        "calibration_scan_reduce_fetch_ratio": round(red * 1024 / (4.0 * max(red_items, 1)), 4),
        "k_fill": {
            "kernel": fill,
            "hbm_bytes_per_launch": kernels[fill]["hbm_bytes_per_launch"],
            "fetch_bytes_corrected": kernels[fill]["fetch_bytes_corrected"],
            "fetch_bytes_raw": kernels[fill]["fetch_bytes_raw"],
            "narrow_read_note": "the x2 correction is calibrated on requests that move a whole 128-byte line (wide streaming reads; 128-byte "
                                "cells read by 32 lanes: raw 66 B per request, profiles/r03_gather_gran_b.json).  A request for a 32- or 64-byte "
                                "cell (BASELINE configs 4 / 5: short buckets) is tallied at the same 66 B, and the microbenchmark's rates say such "
                                "a request moves less than a line (38.8 G 64-byte cells/s = 5.0 TB/s if lines moved, against 3.9 TB/s for 128-byte "
                                "cells): for kernels whose reads are cell gathers the fabric bytes lie between fetch_bytes_raw and fetch_bytes_corrected",
            "write_bytes": kernels[fill]["write_bytes"],
            "algorithmic_bytes_per_launch": int(bench["roofline"]["algorithmic_bytes_per_launch"]),
            "avg_launch_ms_kernel_trace": kernels[fill]["avg_launch_ms_kernel_trace"],
            "avg_launch_ms_bench_hip_events": bench["roofline"]["avg_launch_ms"],
        },
    }
    for name in (f"{a.round}_pmc_summary_{a.tag}.json", f"pmc_summary_current{sfx}.json"):
        with open(os.path.join(dst, name), "w") as f:
            json.dump(out, f, indent=1)
    print(json.dumps(out["k_fill"], indent=1))
    print("calibration ratio (expect ~0.50):", out["calibration_scan_reduce_fetch_ratio"])


if __name__ == "__main__":
    main()
