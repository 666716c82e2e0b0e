#!/bin/bash
# Run ON the GPU box: per-kernel stats of one bench invocation.  Usage: scripts/kstats.sh <tag> [bench args]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --steps 20 --warmup 3 --no-cpu-baseline "$@" > "$OUT/bench.json" 2> "$OUT/err.txt"
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
for f in glob.glob(os.path.join(sys.argv[1], "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        print("%-70s calls %4s avg %10.1f us  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
