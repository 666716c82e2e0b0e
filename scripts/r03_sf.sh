#!/bin/bash
OUT=gpurun_out/r03c; mkdir -p $OUT
timeout 900 python -m pytest tests/test_tokens_seqfirst.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do for rm in 0 1 5 4; do echo "raw_mode=$rm cfg2sf: $(BSQ_RAW_MODE=$rm python3 bench.py --workload cfg2sf --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('loop %.2f us sustained %.2f us frac %.3f' % (r['kernel_avg_ms']*1e3, j['sustained']['kernel_avg_ms']*1e3, j['sustained']['frac']))")"; done; done | tee $OUT/seqfirst_512.txt
