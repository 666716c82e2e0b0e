#!/bin/bash
# usage: scripts/r03_kstats.sh <tag> <bench args...>: kernel-trace stats of one bench invocation (top kernels, avg us)
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-e2e --no-sustained --steps 20 --warmup 3 "$@" > /dev/null 2> $OUT/err.txt
python3 - <<PY
import csv,glob
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:5]: print("%-70s calls %5s avg %8.2f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
