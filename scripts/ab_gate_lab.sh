#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
one() { BSQ_EXPAND_PAD=$2 timeout 300 python3 bench.py --full-line --workload $1 --no-configs --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('loop %.1f us  sustained %.1f us (frac %.3f)' % (r['kernel_avg_ms']*1e3, d['sustained']['kernel_avg_ms']*1e3, d['sustained']['frac']))"; }
for rep in 1 2; do
  for v in old new; do
    cp ab/$v.so bioseq_amd/libbsq_hip.so
    for W in cfg3 cfg3b cfg4f; do for PAD in 0 24576; do echo "$v $W pad=$PAD: $(one $W $PAD)"; done; done
  done
done
