#!/bin/bash
# Run ON the GPU box: scalar 64-bit reciprocal chunk arithmetic (chunk_math 0) vs the double reciprocals of round 1 (1), interleaved.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
run() { local label=$1; shift
  r=$(env "$@" timeout 300 python3 "$REPO/bench.py" --full-line --workload $W --steps 40 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('avg %.4f med %.4f min %.4f per-step-events %.4f frac %.3f' % (r['kernel_avg_ms'], r['kernel_median_ms'], r['kernel_min_ms'], r['kernel_avg_ms_per_step_events'], r['frac']))")
  echo "$W $label: $r"; }
for rep in 1 2; do
for W in cfg3 cfg4f; do
  for m in 1 0; do run "chunk_math=$m" BSQ_CHUNK_MATH=$m; done
done; done
W=cfg3
for pad in 24576 36864 53248; do for m in 1 0; do run "chunk_math=$m expand_pad=$pad" BSQ_CHUNK_MATH=$m BSQ_EXPAND_PAD=$pad; done; done
