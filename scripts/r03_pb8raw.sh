#!/bin/bash
# k_tokens_pb8_fast as the raw-id pass of the two-pass one-hot: full GPU suite, fuzz, then cfg3 / cfg4f / cfg2sf A/B (tokens_pb8 = 1: k_tokens_raw)
OUT=gpurun_out/r03pb8raw; mkdir -p $OUT
( time timeout 3000 python -m pytest tests -m gpu -q -x ) > $OUT/gputest.txt 2>&1; grep -E "passed|failed|error" $OUT/gputest.txt | tail -2
timeout 900 python tests/fuzz_gpu.py ${1:-240} 99 2>&1 | tail -2 | tee $OUT/fuzz.txt
for i in 1 2; do for w in cfg3 cfg4f cfg2sf; do for v in 1 0; do echo "tokens_pb8=$v $w: $(BSQ_TOKENS_PB8=$v python3 bench.py --workload $w --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('loop %.2f us sustained %.2f us frac %.3f' % (r['kernel_avg_ms']*1e3, j['sustained']['kernel_avg_ms']*1e3, j['sustained']['frac']))")"; done; done; done | tee $OUT/ab.txt
