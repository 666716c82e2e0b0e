#!/bin/bash
# Run ON the GPU box: token pass k_tokens_raw (raw_mode 1) vs k_tokens_raw2 with the LDS table (2) / register table (3 = auto), interleaved.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
run() { local label=$1; shift
  r=$(env "$@" timeout 300 python3 "$REPO/bench.py" --full-line --workload $W --steps 40 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms/step %.4f kernel %.4f min %.4f frac %.3f' % (d['ms_per_step'], r['kernel_avg_ms'], r['kernel_min_ms'], r['frac']))")
  echo "$W $label: $r"; }
for rep in 1 2; do for W in cfg2sf cfg3 cfg4f; do for m in 1 2 3; do run "raw_mode=$m" BSQ_RAW_MODE=$m; done; done; done
