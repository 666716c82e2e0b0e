#!/bin/bash
OUT=gpurun_out/r03d; mkdir -p $OUT
timeout 900 python -m pytest tests/test_augment.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do echo "cfg5aug: $(python3 bench.py --workload cfg5aug --no-cpu-baseline --no-e2e --no-sustained 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('loop %.2f us frac %.3f' % (r['kernel_avg_ms']*1e3, r['frac']))")"; done | tee $OUT/augment_overlap.txt
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/augtrace -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg5aug --no-cpu-baseline --no-e2e --no-sustained --steps 20 --warmup 3 > /dev/null 2>&1; python3 - <<PY
import csv,glob
for f in glob.glob("$GRAFT_REPO_ROOT/$OUT/augtrace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:3]: print(r["Name"][:60], r["Calls"], "avg us", float(r["AverageNs"])/1e3)
PY
