#!/bin/bash
# Run ON the GPU box: chunk class = blockIdx % 8 (default) vs class = HW_REG_XCC_ID + per-class atomic slot counters (xcd_claim=1), interleaved.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
run() { local label=$1; shift
  r=$(env "$@" timeout 300 python3 "$REPO/bench.py" --full-line --workload $W --steps 40 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms/step %.4f kernel %.4f min %.4f frac %.3f' % (d['ms_per_step'], r['kernel_avg_ms'], r['kernel_min_ms'], r['frac']))")
  echo "$W $label: $r"; }
for rep in 1 2; do for W in cfg3 cfg4f; do for c in 0 1; do run "xcd_claim=$c" BSQ_XCD_CLAIM=$c; done; done; done
