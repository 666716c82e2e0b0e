#!/bin/bash
# Run ON the GPU box: FETCH_SIZE and SQ counters per kernel (each rocprofv3 pass under its own timeout:
# a counter set the hardware cannot collect makes rocprofv3 abort and then hang in its finaliser) for one bench invocation.  Usage: pmc_fetch.sh <tag> [bench args]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --no-cpu-baseline $*"
timeout 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/a" -- python3 "$REPO/bench.py" $ARGS > /dev/null 2> "$OUT/a.err"
timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/b" -- python3 "$REPO/bench.py" $ARGS > /dev/null 2> "$OUT/b.err"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
for sub in ("a", "b"):
    for f in glob.glob(os.path.join(sys.argv[1], sub, "**", "*counter_collection.csv"), recursive=True):
        d = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            d[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in d.items():
            if "k_onehot" in k or "k_expand" in k or "k_tokens" in k or "k_tokenize" in k:
                print(k[:60], " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
PY
