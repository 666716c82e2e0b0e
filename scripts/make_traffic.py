#!/usr/bin/env python3
"""Copy the per-workload rocprofv3 summaries from gpurun_out/<tag>_<workload>/ (scripts/gpu_evidence.sh profile) into
profiles/<tag>/ and rebuild profiles/traffic.json.

HBM bytes per step = sum over the step's kernels of WRITE_SIZE + 2 x FETCH_SIZE (both in KB, separate --pmc passes; FETCH_SIZE
doubled as MI355X_MICROARCH.md prescribes for gfx950).  Kernel times are those of the K TIMED steps only (summarize_prof.py:
kernel_timed_avg_us, kernel_median_us, n_timed), so that algorithmic bytes / SUM kernel_timed_avg_us / 8 TB/s reproduces the
`frac` that same profiled run printed (`frac_from_trace` vs `frac_printed_under_rocprof`)."""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
dst = os.path.join(ROOT, "profiles", tag)
os.makedirs(dst, exist_ok=True)
traffic = {"_note": "HBM bytes per step from separate rocprofv3 --pmc passes (WRITE_SIZE, FETCH_SIZE; units KB), summed over the "
                    "kernels of one bench step (the K timed steps only). FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for "
                    "gfx950 (an upper bound). kernel_timed_avg_us: per step, over the K timed steps of the profiled run. Sources: "
                    "profiles/%s/<workload>_summary.json (scripts/gpu_evidence.sh profile + scripts/make_traffic.py)." % tag}
old = {}
try:
    old = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
except Exception:
    pass
STEP_KERNELS = ("k_augment_tokens_fused", "k_tokens_pb8", "k_tokens_raw", "k_expand_chunks", "k_expand_rows1", "k_expand_small", "k_expand_bcl", "k_onehot_tile",
                "k_onehot_chunks", "k_tokenize_chunks", "k_tokens_bp8", "k_augment", "k_tokenize_rows", "k_tokenize_tile", "k_onehot_rows")


def short(name):
    for k in STEP_KERNELS:
        if k in name:
            return k
    return None


for w in ("cfg3", "cfg3b", "cfg2", "cfg2sf", "cfg5", "cfg5aug", "cfg4f", "cfg4b", "cfg3bcl", "cfg1oh"):
    src = os.path.join(ROOT, "gpurun_out", "%s_%s" % (tag, w))
    sj = os.path.join(src, "summary.json")
    if not os.path.exists(sj):
        if w in old:
            traffic[w] = old[w]  # (an earlier round's entry stays until it is re-measured)
        continue
    shutil.copy(sj, os.path.join(dst, "%s_summary.json" % w))
    shutil.copy(os.path.join(src, "summary.txt"), os.path.join(dst, "%s_summary.txt" % w))
    if os.path.exists(os.path.join(src, "kernel_stats.csv")):
        shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(dst, "%s_kernel_stats.csv" % w))
    bt = os.path.join(src, "bench_trace.json")
    if os.path.exists(bt):
        shutil.copy(bt, os.path.join(dst, "%s_bench_under_rocprof.json" % w))
    s = json.load(open(sj))
    entry = {"WRITE_SIZE_KB": {}, "FETCH_SIZE_KB": {}, "kernel_timed_avg_us": {}, "kernel_median_us": {}, "n_timed": {}, "source": "profiles/%s" % tag}
    for ctr in ("WRITE_SIZE", "FETCH_SIZE"):
        for name, v in s.get(ctr, {}).items():
            k = short(name)
            if k and "timed_avg_per_step" in v:  # step kernels only (the untimed check's kernels are not)
                entry[ctr + "_KB"][k] = entry[ctr + "_KB"].get(k, 0.0) + v["timed_avg_per_step"]
    for name, v in s.get("kernel_trace", {}).items():
        k = short(name)
        if k and "kernel_timed_avg_us" in v:
            entry["kernel_timed_avg_us"][k] = entry["kernel_timed_avg_us"].get(k, 0.0) + v["kernel_timed_avg_us"]
            entry["kernel_median_us"][k] = v["kernel_median_us"]
            entry["n_timed"][k] = v["n_timed"]
    wr, fe = sum(entry["WRITE_SIZE_KB"].values()), sum(entry["FETCH_SIZE_KB"].values())
    if wr:
        entry["hbm_bytes_per_launch"] = int((wr + 2 * fe) * 1024)
        entry["hbm_bytes_per_launch_fetch_undoubled"] = int((wr + fe) * 1024)
    try:  # the build the counters were measured on (bench.py prints bsq_build_id(); it flags `traffic_stale` when the loaded library differs)
        entry["build_id"] = json.loads(open(bt).read().strip().splitlines()[-1]).get("build_id")
    except Exception:
        entry["build_id"] = None
    rc = s.get("roofline_check")
    if rc:
        entry["algorithmic_bytes_per_launch"] = rc["algorithmic_bytes_per_launch"]
        entry["frac_from_trace"] = rc["frac_from_trace"]
        entry["frac_printed_under_rocprof"] = rc["frac_printed"]
        if wr:
            entry["traffic_over_algorithmic"] = entry["hbm_bytes_per_launch"] / rc["algorithmic_bytes_per_launch"]
    # the UN-profiled line of the same build on the same box (scripts/gpu_evidence.sh bench).  rocprofv3 serialises dispatches (a gap of
    # ~10 us around each), so an event pair around K launches of a 16-40 us kernel reads far too long UNDER the profiler
    # (frac_printed_under_rocprof); the kernel durations of the trace are the ones to hold against the un-profiled event times.
    ub = os.path.join(ROOT, "gpurun_out", tag, "bench_%s.json" % w)
    if os.path.exists(ub):
        try:
            j = json.loads(open(ub).read().strip().splitlines()[-1])
            entry["frac_unprofiled_loop"] = j["roofline"]["frac"]
            entry["frac_unprofiled_sustained"] = (j.get("sustained") or {}).get("frac")
            if j.get("cold"):
                entry["cold"] = {k: j["cold"][k] for k in ("batches", "input_bytes_total", "ms_per_step", "frac", "sustained_ms_per_step", "frac_sustained", "copy_mix_ms", "frac_of_copy_mix") if k in j["cold"]}
            shutil.copy(ub, os.path.join(dst, "bench_%s.json" % w))
        except Exception:
            pass
    traffic[w] = entry
    print(w, json.dumps(entry))
# the round's other records: the driver's line, bench lines, GPU suite, harness, SQ / TCC counter summaries
for name in ("bench_default.json", "bench_lines.txt", "gputest.txt", "fuzz.txt"):
    f = os.path.join(ROOT, "gpurun_out", tag, name)
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, name))
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag + "_sq*_*"))):
    w = os.path.basename(d).split("_", 2)[2]
    kind = "cold_sq_tcc_counters" if "_sqcold_" in os.path.basename(d) else "sq_tcc_counters"
    if os.path.exists(os.path.join(d, "summary.txt")):
        shutil.copy(os.path.join(d, "summary.txt"), os.path.join(dst, "%s_%s.txt" % (w, kind)))
# The COLD regime is the token workloads' primary `frac` since round 5: its kernel durations from the rocprofv3 kernel trace of
# `bench.py --workload W --cold` (COLD=1 scripts/gpu_evidence.sh sq; the first lines of W_cold_sq_tcc_counters.txt: tens of thousands of
# cold dispatches, a few hundred cache-resident ones among them) beside the event-timed figure of the un-profiled line.
for w, entry in traffic.items():
    f = os.path.join(dst, "%s_cold_sq_tcc_counters.txt" % w)
    if not isinstance(entry, dict) or "cold" not in entry or not os.path.exists(f):
        continue
    found = []  # the step's kernels (one, or raw pass + expansion): lines "name  calls N avg X us" at the head of the file
    for line in open(f):
        k = short(line)
        if k and " calls " in line and " avg " in line:
            found.append((k, int(line.split(" calls ")[1].split()[0]), float(line.split(" avg ")[1].split()[0])))
    if found:
        top = max(c for _, c, _ in found)
        step = [(k, c, a) for k, c, a in found if c >= 0.5 * top]  # (kernels of the untimed check run a handful of times)
        avg_us = sum(a for _, _, a in step)
        entry["cold"]["kernel"] = "+".join(k for k, _, _ in step)
        entry["cold"]["kernel_avg_us_from_trace"] = avg_us
        entry["cold"]["trace_calls"] = top
        entry["cold"]["frac_from_trace"] = entry["algorithmic_bytes_per_launch"] / (avg_us * 1e-6) / 8e12
        entry["cold"]["source"] = "profiles/%s/%s_cold_sq_tcc_counters.txt" % (tag, w)
        print(w, "cold:", json.dumps(entry["cold"]))
json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
