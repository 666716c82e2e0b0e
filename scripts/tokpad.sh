#!/bin/bash
# Run ON the GPU box: bench the token / channels-first kernels with k_tokenize_chunks under occupancy caps (knob tokenize_pad).
for w in cfg2 cfg5 cfg3bcl; do for pad in 0 10240 22528 36864 60000; do
echo -n "$w pad=$pad: "; BSQ_TOKENIZE_PAD=$pad BSQ_BENCH_SKIP_SANITY=1 python bench.py --full-line --workload $w --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['achieved'], d['roofline'].get('kernel_avg_ms'))"
done; done
