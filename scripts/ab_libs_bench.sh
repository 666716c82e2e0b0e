#!/bin/bash
# Run ON the GPU box: ab/old.so vs ab/new.so on bench workloads (arguments), interleaved, 3 repetitions.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
one() { timeout 300 python3 bench.py --full-line --workload $1 --steps 60 --warmup 30 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms/step %.4f kernel %.4f min %.4f frac %.3f' % (d['ms_per_step'], r['kernel_avg_ms'], r['kernel_min_ms'], r['frac']))"; }
for rep in 1 2 3; do
  for v in old new; do
    cp ab/$v.so bioseq_amd/libbsq_hip.so
    for W in "$@"; do echo "$v $W: $(one $W)"; done
  done
done
cp ab/new.so bioseq_amd/libbsq_hip.so
