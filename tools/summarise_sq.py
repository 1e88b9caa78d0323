#!/usr/bin/env python3
"""gpurun_out/sq_<tag>_cfg<N>/ (tools/profile_sq.sh) -> profiles/<round>_sq_<tag>_cfg<N>.json: per kmx kernel the mean of every
SQ counter collected, plus the ratios the 'bound' claims rest on (share of wave cycles spent waiting / issuing, LDS share of
the issued instructions, bank-conflict cycles per LDS instruction)."""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--config", default="3")
    ap.add_argument("--round", default="r04")
    a = ap.parse_args()
    src = os.path.join(ROOT, "gpurun_out", f"sq_{a.tag}_cfg{a.config}")
    per = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(src, "pass*", "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if "kmx::" in row["Kernel_Name"]:
                    per[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {"source": f"tools/profile_sq.sh {a.tag} {a.config}: rocprofv3 --pmc (SQ block, two passes) over `python3 bench.py --config {a.config} "
                     "--no-cpu-baseline --no-open-compare --steps 4 --warmup 1`, MI355X; values are means per launch, SQ cycle counters in quad-cycles "
                     "summed over all waves", "kernels": {}}
    for k, cs in sorted(per.items()):
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        d = {"launches": max(len(v) for v in cs.values()), "counters": {c: round(x, 1) for c, x in sorted(m.items())}}
        wc = m.get("SQ_WAVE_CYCLES")
        if wc:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                if c in m:
                    d[f"{c}_share_of_wave_cycles"] = round(m[c] / wc, 4)
        ai = m.get("SQ_ACTIVE_INST_LDS", 0) + m.get("SQ_ACTIVE_INST_VALU", 0) + m.get("SQ_ACTIVE_INST_VMEM", 0) + m.get("SQ_ACTIVE_INST_SCA", 0)
        if ai:
            for c in ("SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA"):
                if c in m:
                    d[f"{c}_share_of_issue_cycles"] = round(m[c] / ai, 4)
        if m.get("SQ_INSTS_LDS") and "SQ_LDS_BANK_CONFLICT" in m:
            d["bank_conflict_cycles_per_lds_instruction"] = round(m["SQ_LDS_BANK_CONFLICT"] / m["SQ_INSTS_LDS"], 3)
        out["kernels"][k] = d
    dst = os.path.join(ROOT, "profiles", f"{a.round}_sq_{a.tag}_cfg{a.config}.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    for k, d in out["kernels"].items():
        if "validate" in k or "lookup" in k or "fill" in k:
            print(k, json.dumps({x: y for x, y in d.items() if x != "counters"}))


if __name__ == "__main__":
    main()
