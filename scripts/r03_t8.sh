#!/bin/bash
OUT=gpurun_out/r03a; mkdir -p $OUT
python -m pytest tests/test_tokens8.py tests/test_ragged_chunks.py tests/test_bcl_and_loaders.py -m gpu -x -q 2>&1 | tail -5 > $OUT/t8_tests.txt
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cfg2 or cfg4_cfg5 or all_keys or shapes" 2>&1 | tail -5 >> $OUT/t8_tests.txt
python scripts/mix_lab.py > $OUT/mix_lab2.txt 2>&1
for i in 1 2; do for f in 0 1; do for w in cfg2 cfg5; do echo "tokens8_fast=$f $w: $(BSQ_TOKENS8_FAST=$f python3 bench.py --workload $w --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('loop %.2f us  per-step-events avg %.2f min %.2f med %.2f  frac %.3f' % (r['kernel_avg_ms']*1e3, r['kernel_avg_ms_per_step_events']*1e3, r['kernel_min_ms']*1e3, r['kernel_median_ms']*1e3, r['frac']))")"; done; done; done > $OUT/t8_ab.txt
cat $OUT/t8_tests.txt $OUT/t8_ab.txt; grep "k_tokens_bp8\|mode 1 nt (res" $OUT/mix_lab2.txt
