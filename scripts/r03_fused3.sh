#!/bin/bash
OUT=gpurun_out/r03h; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_augment.py tests/test_tokens8.py -m gpu -x -q 2>&1 | tail -8
for i in 1 2; do for f in 0 1; do echo "augment_fused=$f cfg5aug: $(BSQ_AUGMENT_FUSED=$f python3 bench.py --workload cfg5aug --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('loop %.2f us frac %.3f | sustained %.2f us frac %.3f' % (r['kernel_avg_ms']*1e3, r['frac'], j['sustained']['kernel_avg_ms']*1e3, j['sustained']['frac']))")"; done; done | tee $OUT/augment_inwave_ab.txt
for w in cfg5 cfg2; do echo "$w: $(python3 bench.py --workload $w --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('loop %.2f us frac %.3f | sustained %.2f' % (r['kernel_avg_ms']*1e3, r['frac'], j['sustained']['kernel_avg_ms']*1e3))")"; done | tee -a $OUT/augment_inwave_ab.txt
