#!/bin/bash
OUT=gpurun_out/r03pb8b; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_tokens_seqfirst.py -m gpu -x -q 2>&1 | tail -3 | tee $OUT/tests.txt
for i in 1 2 3; do for v in "TOKENS_PB8=1" "PB8_TILE=3" "PB8_TILE=2" "PB8_TILE=2 BSQ_TOKENS8_LOOKUP=1"; do echo "$v cfg2sf: $(env BSQ_$v python3 bench.py --workload cfg2sf --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('loop %.2f us sustained %.2f us frac %.3f' % (r['kernel_avg_ms']*1e3, j['sustained']['kernel_avg_ms']*1e3, j['sustained']['frac']))")"; done; done | tee $OUT/ab.txt
