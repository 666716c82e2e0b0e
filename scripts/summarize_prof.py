#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a small text/JSON summary."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
summary = {}


def find(sub, pat):
    return sorted(glob.glob(os.path.join(out, sub, "**", pat), recursive=True))


for f in find("trace", "*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    summary["kernel_stats"] = [{k: r[k] for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")} for r in rows[:12]]
for f in find("trace", "*kernel_trace.csv"):
    d = defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        d[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        meta[n] = {k: r.get(k) for k in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")}
    summary["kernel_trace"] = {n: dict(calls=len(v), avg_ns=sum(v) / len(v), min_ns=min(v), max_ns=max(v), **meta[n]) for n, v in d.items()}
for sub, ctr in (("pmc_write", "WRITE_SIZE"), ("pmc_fetch", "FETCH_SIZE")):
    for f in find(sub, "*counter_collection.csv"):
        d = defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == ctr:
                d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        summary[ctr] = {n: dict(calls=len(v), avg=sum(v) / len(v), min=min(v), max=max(v)) for n, v in d.items()}
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
for k, v in summary.items():
    print("==", k)
    if isinstance(v, list):
        for r in v:
            print("  ", r)
    else:
        for n, r in v.items():
            print("  ", n[:90], r)
