#!/usr/bin/env python3
"""gpurun_out/subk_<tag>/ (tools/profile_subk.sh) -> profiles/<round>_subk_<tag>.json: per sub-k kernel and probe case the launch
duration (rocprofv3 kernel trace is per kernel over all cases, so durations come from the counter passes' own launch order is not
available: the summary keeps the kernel-trace averages) and the HBM bytes per launch from the PMC passes (FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for gfx950; WRITE_SIZE as reported), largest launches first — one line per (kernel, launch size class)."""
import argparse
import csv
import glob
import json
import os
import re
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    return re.sub(r"\(.*$", "", re.sub(r"^void ", "", name))


def counters(path, name):
    per = defaultdict(list)
    for p in glob.glob(path, recursive=True):
        with open(p, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == name and "k_prefix" in row["Kernel_Name"]:
                    per[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--round", default="r04")
    a = ap.parse_args()
    src = os.path.join(ROOT, "gpurun_out", f"subk_{a.tag}")
    fetch = counters(os.path.join(src, "fetch", "**", "*counter_collection.csv"), "FETCH_SIZE")
    write = counters(os.path.join(src, "write", "**", "*counter_collection.csv"), "WRITE_SIZE")
    avg = {}
    for p in glob.glob(os.path.join(src, "kt", "**", "*kernel_stats.csv"), recursive=True):
        with open(p, newline="") as f:
            for row in csv.DictReader(f):
                if "k_prefix" in row["Name"]:
                    avg[short(row["Name"])] = {"calls": int(row["Calls"]), "avg_ms": round(float(row["AverageNs"]) / 1e6, 4), "max_ms": round(float(row["MaxNs"]) / 1e6, 4)}
    out = {"source": "tools/profile_subk.sh: rocprofv3 --kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes) over "
                     "`python3 tools/probe_prefix.py 7 6 5 3` (DNA4 k=10, 1e8 letters, default 2 prefix levels; 2e5 / 5e4 / 1e4 / 300 queries per launch), MI355X",
           "units": "bytes per launch; fetch = FETCH_SIZE KiB x 1024 x 2 (gfx950 correction), write = WRITE_SIZE KiB x 1024; the launches of a kernel are "
                    "listed largest first (a kernel runs in several probe cases: the probe log says which case launches which kernel)",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        f_ = sorted(fetch.get(k, []), reverse=True)
        w_ = sorted(write.get(k, []), reverse=True)
        classes = []
        # group launches whose counters agree within 2 % (the same probe case repeated)
        for vals, key in ((f_, "fetch_bytes"), (w_, "write_bytes")):
            groups = []
            for v in vals:
                if groups and abs(groups[-1][0] - v) <= 0.02 * max(groups[-1][0], 1.0):
                    groups[-1].append(v)
                else:
                    groups.append([v])
            classes.append([{key: int(sum(g) / len(g) * 1024 * (2 if key == "fetch_bytes" else 1)), "launches": len(g)} for g in groups if sum(g) / len(g) > 16])
        out["kernels"][k] = {"kernel_trace": avg.get(k), "fetch_classes": classes[0][:6], "write_classes": classes[1][:6]}
    dst = os.path.join(ROOT, "profiles", f"{a.round}_subk_{a.tag}.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    for k, v in out["kernels"].items():
        print(k, v["kernel_trace"], v["fetch_classes"][:3], v["write_classes"][:3])


if __name__ == "__main__":
    main()
