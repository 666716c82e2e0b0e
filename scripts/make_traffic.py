#!/usr/bin/env python3
"""Copy the per-workload rocprofv3 summaries from gpurun_out/<tag>_<workload>/ into profiles/<tag>/ and rebuild
profiles/traffic.json: HBM bytes per step = sum over the step's kernels of WRITE_SIZE + 2 x FETCH_SIZE (both in KB,
separate --pmc passes; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950)."""
import json, os, shutil, sys, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
dst = os.path.join(ROOT, "profiles", tag)
os.makedirs(dst, exist_ok=True)
traffic = {"_note": "HBM bytes per step from separate rocprofv3 --pmc passes (WRITE_SIZE, FETCH_SIZE; units KB), summed over the "
                    "kernels of one bench step. FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (an upper "
                    "bound). Sources: profiles/%s/<workload>_summary.json (scripts/evidence_all.sh + scripts/make_traffic.py)." % tag}
STEP_KERNELS = ("k_augment_tokens_fused", "k_tokens_pb8", "k_tokens_raw", "k_expand_chunks", "k_expand_small", "k_expand_bcl", "k_onehot_tile", "k_onehot_chunks", "k_tokenize_chunks", "k_tokens_bp8",
                "k_augment", "k_tokenize_rows", "k_tokenize_tile", "k_onehot_rows")
def short(name):
    for k in STEP_KERNELS:
        if k in name:
            return k
    return None
for w in ("cfg3", "cfg2", "cfg5", "cfg4f", "cfg4b", "cfg5aug", "cfg3bcl", "cfg2sf", "cfg1oh"):
    src = os.path.join(ROOT, "gpurun_out", "%s_%s" % (tag, w))
    sj = os.path.join(src, "summary.json")
    if not os.path.exists(sj):
        continue
    shutil.copy(sj, os.path.join(dst, "%s_summary.json" % w))
    shutil.copy(os.path.join(src, "summary.txt"), os.path.join(dst, "%s_summary.txt" % w))
    for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(dst, "%s_kernel_stats.csv" % w))
    bt = os.path.join(src, "bench_trace.json")
    if os.path.exists(bt):
        shutil.copy(bt, os.path.join(dst, "%s_bench_under_rocprof.json" % w))
    s = json.load(open(sj))
    entry = {"WRITE_SIZE_KB": {}, "FETCH_SIZE_KB": {}, "kernel_avg_us": {}}
    algo = None
    try:
        algo = json.loads(open(bt).read().strip().splitlines()[-1])["roofline"]["algorithmic_bytes_per_launch"]
    except Exception:
        pass
    for ctr in ("WRITE_SIZE", "FETCH_SIZE"):
        for name, v in s.get(ctr, {}).items():
            k = short(name)
            if k:
                entry[ctr + "_KB"][k] = entry[ctr + "_KB"].get(k, 0.0) + v["avg"]
    for name, v in s.get("kernel_trace", {}).items():
        k = short(name)
        if k:
            entry["kernel_avg_us"][k] = v["avg_ns"] / 1e3
    if w == "cfg5aug" and "k_augment_tokens_fused" in entry["WRITE_SIZE_KB"]:
        # the step is ONE launch; bench.py's untimed correctness check tokenises the mutated batch once more with k_tokens_bp8_fast
        for ctr in ("WRITE_SIZE_KB", "FETCH_SIZE_KB", "kernel_avg_us"):
            entry[ctr].pop("k_tokens_bp8", None)
    wr, fe = sum(entry["WRITE_SIZE_KB"].values()), sum(entry["FETCH_SIZE_KB"].values())
    if wr:
        entry["hbm_bytes_per_launch"] = int((wr + 2 * fe) * 1024)
        entry["hbm_bytes_per_launch_fetch_undoubled"] = int((wr + fe) * 1024)
    if algo:
        entry["algorithmic_bytes_per_launch"] = algo
    traffic[w] = entry
    print(w, entry)
json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
