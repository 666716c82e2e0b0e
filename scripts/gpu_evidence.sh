#!/bin/bash
# Run ON the GPU box (through gpurun).  One script for every measurement pass of a round (round 3 had 23 one-off files):
#
#   scripts/gpu_evidence.sh tests [pytest args]        GPU suite -> gpurun_out/<TAG>/gputest.txt
#   scripts/gpu_evidence.sh fuzz [seconds] [seed]       randomised harness -> gpurun_out/<TAG>/fuzz.txt
#   scripts/gpu_evidence.sh bench                       the driver's line (python3 bench.py) + one line per workload -> bench_*.json, bench_lines.txt
#   scripts/gpu_evidence.sh profile [workloads...]      per workload: rocprofv3 --kernel-trace --stats, then SEPARATE --pmc WRITE_SIZE and
#                                                       --pmc FETCH_SIZE passes of `bench.py --workload W --no-configs`, condensed by
#                                                       summarize_prof.py -> gpurun_out/<TAG>_<W>/summary.{txt,json}
#   scripts/gpu_evidence.sh sq [workloads...]           SQ / TCC counter passes -> gpurun_out/<TAG>_sq_<W>/summary.txt
#   scripts/gpu_evidence.sh dispatch                    scripts/check_dispatch.py over scripts/dispatch_shapes.json (the automatic one-hot path against every
#                                                       forced path, 5 % gate) -> gpurun_out/<TAG>/dispatch_check.txt
#   scripts/gpu_evidence.sh all                         tests, fuzz 300, bench, dispatch, profile + sq of every workload
#
# TAG (environment, default r04) names the output directories.  Back in the build container:
#   python scripts/make_traffic.py <TAG>   copies the summaries into profiles/<TAG>/ and rebuilds profiles/traffic.json.
# Every rocprofv3 command runs `python3 bench.py ...` directly (no env / bash -c hop) under its own timeout; counters are never combined
# with a trace domain.
set -u
TAG=${TAG:-r06}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
ALL="cfg3 cfg3b cfg2 cfg2sf cfg5 cfg5aug cfg4f cfg4b cfg3bcl"
MODE=${1:-all}; shift || true
OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp

do_tests() { ( cd "$REPO" && time timeout 3000 python3 -m pytest tests -m gpu -q "$@" ) > "$OUT/gputest.txt" 2>&1; grep -E "passed|failed|error" "$OUT/gputest.txt" | tail -2; }
do_fuzz() { ( cd "$REPO" && timeout $(( ${1:-300} + 600 )) python3 tests/fuzz_gpu.py "${1:-300}" "${2:-77}" ) 2>&1 | tail -2 | tee "$OUT/fuzz.txt"; }
do_bench() {
  cd "$REPO"
  # exactly the driver's command: its stdout is the compact <= 4-KB line; the full object lands in bench_full.json beside bench.py
  ( time python3 bench.py ) > "$OUT/bench_default_line.json" 2> "$OUT/bench_default.err"
  cp "$REPO/bench_full.json" "$OUT/bench_default.json"
  wc -c "$OUT/bench_default_line.json"
  for w in $ALL; do python3 bench.py --full-line --workload $w --no-configs --no-cpu-baseline --cold > "$OUT/bench_$w.json" 2>> "$OUT/bench.err"; done
  python3 - "$OUT" $ALL <<'PY' | tee "$OUT/bench_lines.txt"
import json, sys
out = sys.argv[1]
for w in sys.argv[2:]:
    try:
        j = json.loads(open("%s/bench_%s.json" % (out, w)).read().strip().splitlines()[-1])
    except Exception as ex:
        print("%-8s FAILED %r" % (w, ex)); continue
    r, s, c = j["roofline"], j.get("sustained") or {}, j.get("cold") or {}
    print("%-8s loop %.4f ms frac %.3f | sustained %.4f ms frac %.3f | cold %s | of_fill %.3f of_mix %s | check %s" % (
        w, r["kernel_avg_ms"], r["frac"], s.get("kernel_avg_ms", 0), s.get("frac", 0),
        ("%.4f ms frac %.3f (sustained %.3f, of_mix %.3f)" % (c["ms_per_step"], c["frac"], c["frac_sustained"], c.get("frac_of_copy_mix") or 0)) if c else "-",
        r["frac_of_fill"], r["frac_of_copy_mix"] and round(r["frac_of_copy_mix"], 3), j["check"].get("ok")))
d = json.load(open(out + "/bench_default.json"))  # (bench_full.json: the whole object, indented)
print("driver line: cfg3 ms/step %.4f frac %.3f; configs:" % (d["ms_per_step"], d["roofline"]["frac"]))
for w, c in (d.get("configs") or {}).items():
    print("   %-8s %.4f ms frac %.3f sustained %.3f cold %s check %s (%.1f s)" % (w, c["ms_per_step"], c["frac"], c["frac_sustained"],
          ("%.3f" % c["cold"]["frac_sustained"]) if "cold" in c else "-", c["check"].get("ok"), c["seconds"]))
PY
}
do_profile() {
  for w in "$@"; do
    P=$REPO/gpurun_out/${TAG}_$w; mkdir -p "$P"; cd /tmp
    A="--full-line --workload $w --no-configs --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-sustained"
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/trace" -- python3 "$REPO/bench.py" $A > "$P/bench_trace.json" 2> "$P/trace.err"
    timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$P/pmc_write" -- python3 "$REPO/bench.py" $A > "$P/bench_pmc_write.json" 2> "$P/pmc_write.err"
    timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$P/pmc_fetch" -- python3 "$REPO/bench.py" $A > "$P/bench_pmc_fetch.json" 2> "$P/pmc_fetch.err"
    python3 "$REPO/scripts/summarize_prof.py" "$P" > "$P/summary.txt" 2>&1
    find "$P/trace" -name "*kernel_stats.csv" -exec cp {} "$P/kernel_stats.csv" \;
    rm -rf "$P/trace" "$P/pmc_write" "$P/pmc_fetch"   # (per-dispatch CSVs: tens of MB; gpurun copies back at most 64 MiB)
    echo "$w: $(grep -A1 roofline_check "$P/summary.txt" | tail -1 | cut -c1-240)"
  done
}
do_sq() {
  for w in "$@"; do
    P=$REPO/gpurun_out/${TAG}_sq${COLD:+cold}_$w; mkdir -p "$P"; cd /tmp
    # COLD=1: the cold-input regime (bench.py --cold: thousands of dispatches cycling over > 512 MiB of distinct batches dominate every
    # average; the copy-mix yardstick runs the same way and is listed beside the kernel)
    A="--workload $w --no-configs --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-sustained ${COLD:+--cold}"
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/trace" -- python3 "$REPO/bench.py" $A > /dev/null 2> "$P/trace.err"
    timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$P/sq1" -- python3 "$REPO/bench.py" $A > /dev/null 2> "$P/sq1.err"
    timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM --output-format csv -d "$P/sq2" -- python3 "$REPO/bench.py" $A > /dev/null 2> "$P/sq2.err"
    timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum --output-format csv -d "$P/tcc" -- python3 "$REPO/bench.py" $A > /dev/null 2> "$P/tcc.err"
    timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$P/fetch" -- python3 "$REPO/bench.py" $A > /dev/null 2> "$P/fetch.err"
    timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$P/write" -- python3 "$REPO/bench.py" $A > /dev/null 2> "$P/write.err"
    python3 - "$P" <<'PY' > "$P/summary.txt"
import csv, glob, os, sys
from collections import defaultdict
for f in glob.glob(os.path.join(sys.argv[1], "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        print("%-90s calls %4s avg %10.2f us  %5s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
for sub in ("sq1", "sq2", "tcc", "fetch", "write"):
    for f in glob.glob(os.path.join(sys.argv[1], sub, "**", "*counter_collection.csv"), recursive=True):
        d = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            d[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in d.items():
            if "k_" in k and "fill" not in k and ("copy_mix" not in k or os.environ.get("COLD")):
                print(k[:100])
                for c, v in sorted(cs.items()):
                    print("   %-24s avg %.4g (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
    rm -rf "$P/trace" "$P/sq1" "$P/sq2" "$P/tcc" "$P/fetch" "$P/write"
    echo "sq $w: $(wc -l < "$P/summary.txt") lines"
  done
}
do_dispatch() { ( cd "$REPO" && timeout 1500 python3 scripts/check_dispatch.py --json "$OUT/dispatch_check.json" ) > "$OUT/dispatch_check.txt" 2>> "$OUT/dispatch_check.err"; grep -E "^#|^FAIL" "$OUT/dispatch_check.txt"; }
case "$MODE" in
  dispatch) do_dispatch ;;
  tests) do_tests "$@" ;;
  fuzz) do_fuzz "$@" ;;
  bench) do_bench ;;
  profile) do_profile ${@:-$ALL} ;;
  sq) do_sq ${@:-cfg2 cfg2sf cfg3b cfg5 cfg5aug cfg4b} ;;
  all) do_tests; do_fuzz 300; do_bench; do_dispatch; do_profile $ALL; do_sq cfg2 cfg2sf cfg3b cfg5 cfg5aug cfg4b ;;
  *) echo "unknown mode $MODE"; exit 2 ;;
esac
