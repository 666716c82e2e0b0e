#!/bin/bash
OUT=gpurun_out/r03d; mkdir -p $OUT
timeout 900 python -m pytest tests/test_augment.py tests/test_bcl_and_loaders.py tests/test_index_batches.py tests/test_every_device.py -m gpu -x -q 2>&1 | tail -5
for i in 1 2; do for ap in 0 1; do echo "augment_path=$ap cfg5aug: $(BSQ_AUGMENT_PATH=$ap python3 bench.py --workload cfg5aug --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('loop %.2f us frac %.3f | sustained %.2f us frac %.3f' % (r['kernel_avg_ms']*1e3, r['frac'], j['sustained']['kernel_avg_ms']*1e3, j['sustained']['frac']))")"; done; done | tee $OUT/augment_staged_ab.txt
