#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
run() { local label=$1; shift
  r=$(env "$@" timeout 300 python3 "$REPO/bench.py" --full-line --workload $W --steps 40 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms/step %.4f kernel %.4f min %.4f frac %.3f' % (d['ms_per_step'], r['kernel_avg_ms'], r['kernel_min_ms'], r['frac']))")
  echo "$W $label: $r"; }
W=cfg3bcl
for rep in 1 2; do run "bcl_path=1 (single pass)" BSQ_BCL_PATH=1; for pad in 0 36864 24576 -1; do run "bcl_path=2 bcl_pad=$pad" BSQ_BCL_PATH=2 BSQ_BCL_PAD=$pad; done; done
