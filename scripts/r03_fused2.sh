#!/bin/bash
OUT=gpurun_out/r03g; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_augment.py -m gpu -x -q 2>&1 | tail -4
for i in 1 2; do for f in 0 1; do for k in 1 2 4; do echo "augment_fused=$f augment_k=$k cfg5aug: $(BSQ_AUGMENT_FUSED=$f BSQ_AUGMENT_K=$k python3 bench.py --workload cfg5aug --no-cpu-baseline --no-e2e --no-sustained 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('loop %.2f us frac %.3f' % (r['kernel_avg_ms']*1e3, r['frac']))")"; done; done; done | tee $OUT/augment_fused_k_ab.txt
bash scripts/r03_kstats.sh r03g/ks_fused_k4 --workload cfg5aug | tee -a $OUT/augment_fused_k_ab.txt
BSQ_AUGMENT_FUSED=1 bash scripts/r03_kstats.sh r03g/ks_seq_k4 --workload cfg5aug | tee -a $OUT/augment_fused_k_ab.txt
