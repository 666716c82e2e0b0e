#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
run() { local label=$1; shift
  r=$(env "$@" timeout 300 python3 "$REPO/bench.py" --full-line --workload $W --steps 40 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('avg %.4f med %.4f min %.4f frac %.3f' % (r['kernel_avg_ms'], r['kernel_median_ms'], r['kernel_min_ms'], r['frac']))")
  echo "$W $label: $r"; }
for W in cfg4b cfg4f; do
for rep in 1 2; do for g in 1 4 16 32 64 128 512; do run "tile_group=$g" BSQ_TILE_GROUP=$g; done; done
done
