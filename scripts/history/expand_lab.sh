#!/bin/bash
# Run ON the GPU box: k_expand_chunks / k_expand_small variants on the small-row shapes (two-pass path forced).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
run() { # label, env...
  local label=$1; shift
  r=$(env "$@" timeout 300 python3 "$REPO/bench.py" --full-line --workload $W --steps 30 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms %.4f min %.4f frac %.3f' % (r['kernel_avg_ms'], r['kernel_min_ms'], r['frac']))")
  echo "$W $label: $r"
}
for W in cfg4f cfg4b; do
  run "auto path" BSQ_ONEHOT_PATH=0
  for m in 1 2 3 4; do
    for pad in -1 16384 36864; do
      run "two-pass mode=$m pad=$pad" BSQ_ONEHOT_PATH=2 BSQ_EXPAND_MODE=$m BSQ_EXPAND_PAD=$pad
    done
  done
  run "two-pass mode=9 (no token loads)" BSQ_ONEHOT_PATH=2 BSQ_EXPAND_MODE=9 BSQ_EXPAND_PAD=-1 BSQ_BENCH_SKIP_SANITY=1
done
W=cfg3
run "default" BSQ_ONEHOT_PATH=0
