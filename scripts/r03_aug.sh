#!/bin/bash
OUT=gpurun_out/r03a; mkdir -p $OUT
timeout 900 python -m pytest tests/test_augment.py tests/test_bcl_and_loaders.py tests/test_every_device.py -m gpu -x -q 2>&1 | tail -15 > $OUT/aug_tests.txt
cat $OUT/aug_tests.txt
for i in 1 2; do for w in cfg5aug cfg5; do echo "$w: $(python3 bench.py --workload $w --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('loop %.2f us  per-step-events avg %.2f min %.2f  frac %.3f' % (r['kernel_avg_ms']*1e3, r['kernel_avg_ms_per_step_events']*1e3, r['kernel_min_ms']*1e3, r['frac']))")"; done; done | tee $OUT/aug_bench.txt
for k in 1 2 4; do echo "augment_waves=$k: $(BSQ_AUGMENT_WAVES=$k python3 bench.py --workload cfg5aug --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('loop %.2f us' % (r['kernel_avg_ms']*1e3))")"; done | tee -a $OUT/aug_bench.txt
