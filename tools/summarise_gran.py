#!/usr/bin/env python3
"""gpurun_out/gran_<tag>/ (tools/profile_gran.sh) -> profiles/<round>_gather_gran_<tag>.json: per (cell stride, bytes read per
cell) of the random-read microbenchmark the rate (M cells/s) and the FETCH_SIZE counter per cell, i.e. the calibration of that
counter for NARROW RANDOM reads (MI355X_MICROARCH.md calibrates its x2 correction on wide streaming reads only)."""
import argparse
import csv
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--round", default="r03")
    a = ap.parse_args()
    src = os.path.join(ROOT, "gpurun_out", f"gran_{a.tag}")
    rows = []
    pat = re.compile(r"table (\d+) MiB stride\s+(\d+) mode (\d+) \(\s*(\d+) B/cell\): ([\d.]+) ms\s+(\d+) M cells/s")
    for name in ("timing.log", "timing_100MiB.log"):
        path = os.path.join(src, name)
        if not os.path.exists(path):
            continue
        for line in open(path):
            m = pat.search(line)
            if m:
                rows.append({"table_MiB": int(m.group(1)), "stride": int(m.group(2)), "mode": int(m.group(3)), "bytes_per_cell": int(m.group(4)),
                             "ms": float(m.group(5)), "M_cells_per_s": int(m.group(6))})
    # the counter pass: dispatches in program order, 7 per (stride, mode) combination (2 warm + 5 timed), same order as the log
    fetch = []
    for path in glob.glob(os.path.join(src, "fetch", "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == "FETCH_SIZE" and "k_gather" in row["Kernel_Name"]:   # (k_gather and k_gather_coop, program order)
                    fetch.append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    fetch.sort()
    big = [r for r in rows if r["table_MiB"] >= 1024]
    n_cells = 12_500_000
    if fetch and len(fetch) == 7 * len(big):
        for i, r in enumerate(big):
            vals = [v for _, v in fetch[7 * i + 2:7 * i + 7]]
            kib = sum(vals) / len(vals)
            r["FETCH_SIZE_KiB_per_dispatch"] = round(kib, 1)
            r["FETCH_SIZE_bytes_per_cell"] = round(kib * 1024 / n_cells, 2)
            # what the counter would have to be multiplied with if every cell cost one full 128-B line / exactly the bytes read
            r["factor_if_128B_line_per_cell"] = round(128.0 / (kib * 1024 / n_cells), 3)
            r["factor_if_only_bytes_read"] = round(r["bytes_per_cell"] / (kib * 1024 / n_cells), 3)
    out = {"source": f"tools/profile_gran.sh {a.tag}: tools/micro/gather_gran.hip, 12.5e6 random cells of a 4 GiB (and a 100 MiB) table, MI355X; "
                     "timing from HIP events in the binary, FETCH_SIZE from a separate rocprofv3 --pmc pass (raw counter, no correction)",
           "n_cells": n_cells, "dispatch_rows_matched": bool(fetch and len(fetch) == 7 * len(big)), "rows": rows}
    dst = os.path.join(ROOT, "profiles", f"{a.round}_gather_gran_{a.tag}.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(dst)
    for r in rows:
        print(r)


if __name__ == "__main__":
    main()
