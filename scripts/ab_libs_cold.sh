#!/bin/bash
# Run ON the GPU box: ab/old.so vs ab/new.so (two builds of libbsq_hip.so) on bench workloads (arguments), interleaved, 3 repetitions:
# loop, sustained and COLD-regime averages (bench.py --full-line --cold).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
one() { timeout 300 python3 bench.py --full-line --workload $1 --no-configs --no-cpu-baseline --no-e2e --cold 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d.get('cold') or {}
print('loop %.2f us  sustained %.2f us (frac %.3f)  cold %.2f us sustained %.2f us (frac %.3f, copy-mix %.2f us)' % (r['kernel_avg_ms']*1e3, d['sustained']['kernel_avg_ms']*1e3, d['sustained']['frac'], c.get('ms_per_step',0)*1e3, c.get('sustained_ms_per_step',0)*1e3, c.get('frac_sustained',0), c.get('copy_mix_ms',0)*1e3))"; }
for rep in 1 2 3; do
  for v in old new; do
    cp ab/$v.so bioseq_amd/libbsq_hip.so
    for W in "$@"; do echo "$v $W: $(one $W)"; done
  done
done
cp ab/new.so bioseq_amd/libbsq_hip.so
