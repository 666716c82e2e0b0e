#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
run() { local label=$1; shift
  r=$(env "$@" timeout 300 python3 "$REPO/bench.py" --full-line --workload $W --steps 40 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('avg %.4f med %.4f min %.4f per-step-events %.4f frac %.3f' % (r['kernel_avg_ms'], r['kernel_median_ms'], r['kernel_min_ms'], r['kernel_avg_ms_per_step_events'], r['frac']))")
  echo "$W $label: $r"; }
W=cfg4f
for pad in 16384 20480 24576 30720 36864; do
  run "expand_small pad=$pad" BSQ_EXPAND_MODE=2 BSQ_EXPAND_PAD=$pad
  run "expand_chunks pad=$pad" BSQ_EXPAND_MODE=1 BSQ_EXPAND_PAD=$pad
done
W=cfg4b
run "tile (auto)" BSQ_ONEHOT_PATH=0
for pad in -1 8192 16384 24576 36864; do
  run "two-pass expand_small pad=$pad" BSQ_ONEHOT_PATH=2 BSQ_EXPAND_MODE=2 BSQ_EXPAND_PAD=$pad
  run "two-pass expand_chunks pad=$pad" BSQ_ONEHOT_PATH=2 BSQ_EXPAND_MODE=1 BSQ_EXPAND_PAD=$pad
done
W=cfg3
for pad in 0 24576 -1; do run "pad=$pad" BSQ_EXPAND_PAD=$pad; done
