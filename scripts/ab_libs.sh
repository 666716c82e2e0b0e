#!/bin/bash
# Run ON the GPU box: interleaved A/B of two builds of libbsq_hip.so (ab/old.so, ab/new.so: built by hand from two
# source states) on the bench workloads given as arguments and on the (P,B) int8 token shapes.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
one() { timeout 300 python3 bench.py --full-line --workload $1 --steps 40 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms/step %.4f kernel %.4f frac %.3f' % (d['ms_per_step'], r['kernel_avg_ms'], r['frac']))"; }
for rep in 1 2; do
  for v in old new; do
    cp ab/$v.so bioseq_amd/libbsq_hip.so
    for W in "$@"; do echo "$v $W: $(one $W)"; done
    echo "$v seqfirst:"; BSQ_QUIET=1 python3 scripts/seqfirst_lab.py 2>&1 | grep -v amdgpu | awk 'NR%3==0' | cut -c1-62
  done
done
cp ab/new.so bioseq_amd/libbsq_hip.so
