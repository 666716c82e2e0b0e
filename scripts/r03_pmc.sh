#!/bin/bash
# usage: scripts/r03_pmc.sh <tag> <bench args...>   -- kernel stats + SQ counters of one bench invocation (separate passes)
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-sustained $*"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" $ARGS > /dev/null 2> "$OUT/trace.err"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$OUT/sq1" -- python3 "$REPO/bench.py" $ARGS > /dev/null 2> "$OUT/sq1.err"
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM --output-format csv -d "$OUT/sq2" -- python3 "$REPO/bench.py" $ARGS > /dev/null 2> "$OUT/sq2.err"
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum --output-format csv -d "$OUT/tcc" -- python3 "$REPO/bench.py" $ARGS > /dev/null 2> "$OUT/tcc.err"
python3 - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, os, sys
from collections import defaultdict
for f in glob.glob(os.path.join(sys.argv[1], "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        print("%-90s calls %4s avg %10.2f us  %5s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
for sub in ("sq1", "sq2", "tcc"):
    for f in glob.glob(os.path.join(sys.argv[1], sub, "**", "*counter_collection.csv"), recursive=True):
        d = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            d[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in d.items():
            if "k_" in k and "fill" not in k:
                print(k[:100])
                for c, v in sorted(cs.items()):
                    print("   %-24s avg %.4g (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
