#!/bin/bash
# Run ON the GPU box: every bench workload with nt (1) and sc1 (2) output stores.  Usage: scripts/store_kind_sweep.sh <tag>
TAG=${1:-store_kind}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG.txt
: > "$OUT"
for w in cfg3 cfg3bcl cfg4f cfg4b cfg2 cfg5; do
  for k in 1 2; do
    for path in 0; do
      r=$(BSQ_NT_STORES=$k timeout 300 python3 "$REPO/bench.py" --full-line --workload $w --steps 30 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%s ms %.4f min %.4f frac %.3f fill %.0f' % (r['kernel'], r['kernel_avg_ms'], r['kernel_min_ms'], r['frac'], r['fill_yardstick_gbps']))")
      echo "$w nt_stores=$k  $r" | tee -a "$OUT"
    done
  done
done
for k in 1 2; do for path in 1 3; do
  r=$(BSQ_NT_STORES=$k BSQ_ONEHOT_PATH=$path timeout 300 python3 "$REPO/bench.py" --full-line --workload cfg3 --steps 30 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms %.4f min %.4f frac %.3f' % (r['kernel_avg_ms'], r['kernel_min_ms'], r['frac']))")
  echo "cfg3 path=$path nt_stores=$k  $r" | tee -a "$OUT"
done; done
