#!/usr/bin/env python3
"""Condense the rocprofv3 CSV output of scripts/gpu_evidence.sh (kernel trace + separate PMC passes of ONE bench.py workload) into
summary.json / summary.txt.

bench.py prints which of its step() calls were the K TIMED ones (`timed`: first_step_index, steps, step_calls_total).  A kernel that
is dispatched m x step_calls_total times is a step kernel with m launches per step; of its dispatches -- in start order -- only
[first * m, (first + K) * m) are kept for `kernel_timed_avg_us` / `kernel_median_us` / `n_timed`, so that

    algorithmic bytes / SUM(kernel_timed_avg_us of the step's kernels) / 8 TB/s

reproduces the `roofline.frac` the same (profiled) run printed (`frac_from_trace` vs `frac_printed`).  The all-dispatch averages
(cold launches, warm-up, the per-step event pass) are kept beside them as `avg_ns`.  PMC passes: the same selection by dispatch order."""
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


def bench_line(path):
    try:
        return json.loads(open(path).read().strip().splitlines()[-1])
    except Exception:
        return None


def timed_slice(n_calls, timed):
    """dispatch index range of the K timed steps for a kernel with n_calls dispatches, or None if it is not a step kernel"""
    if not timed or not timed.get("step_calls_total"):
        return None
    total = timed["step_calls_total"]
    if n_calls < total or n_calls % total:
        return None
    m = n_calls // total
    return timed["first_step_index"] * m, (timed["first_step_index"] + timed["steps"]) * m, m


bt = bench_line(os.path.join(out, "bench_trace.json"))
timed = (bt or {}).get("timed")
for f in find("trace", "*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    summary["kernel_stats"] = [{k: r[k] for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")} for r in rows[:12]]
for f in find("trace", "*kernel_trace.csv"):
    d = defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        d[n].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        meta[n] = {k: r.get(k) for k in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")}
    kt = {}
    step_us = 0.0
    for n, v in d.items():
        v.sort()
        dur = [x[1] for x in v]
        e = dict(calls=len(dur), avg_ns=sum(dur) / len(dur), min_ns=min(dur), max_ns=max(dur), **meta[n])
        sl = timed_slice(len(dur), timed)
        if sl:
            t = sorted(dur[sl[0]:sl[1]])
            e.update(launches_per_step=sl[2], n_timed=len(t), kernel_timed_avg_us=sum(t) / len(t) / 1e3 * sl[2], kernel_median_us=t[len(t) // 2] / 1e3)
            step_us += e["kernel_timed_avg_us"]
        kt[n] = e
    summary["kernel_trace"] = kt
    if bt and step_us:
        algo = bt["roofline"]["algorithmic_bytes_per_launch"]
        summary["roofline_check"] = {"algorithmic_bytes_per_launch": algo, "step_kernels_timed_us": step_us,
                                     "frac_from_trace": algo / (step_us * 1e-6) / 8e12, "frac_printed": bt["roofline"]["frac"],
                                     "note": "both from the SAME profiled run; un-profiled runs are ~1-3 % faster"}
for sub, ctr in (("pmc_write", "WRITE_SIZE"), ("pmc_fetch", "FETCH_SIZE")):
    bp = bench_line(os.path.join(out, "bench_%s.json" % sub))
    tp = (bp or {}).get("timed")
    for f in find(sub, "*counter_collection.csv"):
        d = defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == ctr:
                d[r["Kernel_Name"]].append((int(r.get("Dispatch_Id") or 0), float(r["Counter_Value"])))
        res = {}
        for n, v in d.items():
            v.sort()
            vals = [x[1] for x in v]
            e = dict(calls=len(vals), avg=sum(vals) / len(vals), min=min(vals), max=max(vals))
            sl = timed_slice(len(vals), tp)
            if sl:
                t = vals[sl[0]:sl[1]]
                e.update(n_timed=len(t), timed_avg_per_step=sum(t) / len(t) * sl[2])
            res[n] = e
        summary[ctr] = res
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
for k, v in summary.items():
    print("==", k)
    if isinstance(v, list):
        for r in v:
            print("  ", r)
    elif k == "roofline_check":
        print("  ", v)
    else:
        for n, r in v.items():
            print("  ", n[:90], r)
