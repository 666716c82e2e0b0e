#!/bin/bash
OUT=gpurun_out/r03d; mkdir -p $OUT
python -m pytest tests/test_tokens8.py tests/test_bcl_and_loaders.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do for w in cfg2 cfg5; do echo "$w: $(python3 bench.py --workload $w --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('loop %.2f us frac %.3f | sustained %.2f us frac %.3f | mix yardstick %.2f us' % (r['kernel_avg_ms']*1e3, r['frac'], j['sustained']['kernel_avg_ms']*1e3, j['sustained']['frac'], r['algorithmic_bytes_per_launch']/r['copy_mix_yardstick_gbps']/1e3))")"; done; done | tee $OUT/t8_rules_kernarg.txt
