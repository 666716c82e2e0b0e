#!/bin/bash
# Run ON the GPU box: extra PMC passes (LDS conflicts, L2 hit/miss, wave counts) for the bench command.
TAG=${1:-pmc_extra}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --no-cpu-baseline $*"
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/sq" -- python3 "$REPO/bench.py" $ARGS > /dev/null 2> "$OUT/sq.err"; echo "sq rc=$?"
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum --output-format csv -d "$OUT/tcc" -- python3 "$REPO/bench.py" $ARGS > /dev/null 2> "$OUT/tcc.err"; echo "tcc rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
for sub in ("sq", "tcc"):
    for f in glob.glob(os.path.join(sys.argv[1], sub, "**", "*counter_collection.csv"), recursive=True):
        d = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            d[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in d.items():
            if "k_onehot" in k or "k_expand" in k or "k_tokens" in k or "k_tokenize" in k:
                print(k[:70])
                for c, v in sorted(cs.items()):
                    print("   %-24s avg %.4g (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
