#!/bin/bash
OUT=gpurun_out/${1:-r03c}; mkdir -p $OUT
( time timeout 3000 python -m pytest tests -m gpu -x -q ) > $OUT/gputest.txt 2>&1
grep -E "passed|failed|error" $OUT/gputest.txt | tail -3
timeout 600 python tests/fuzz_gpu.py ${2:-240} 31 2>&1 | tail -3 | tee $OUT/fuzz.txt
for w in cfg3 cfg2 cfg5 cfg4f cfg4b cfg5aug cfg2sf cfg3bcl; do python3 bench.py --workload $w --no-cpu-baseline --no-e2e > $OUT/bench_$w.json 2>> $OUT/bench.err; python3 -c "
import json; j=json.load(open('$OUT/bench_$w.json')); r=j['roofline']
print('%-8s ms/step %.4f  loop %.4f  frac %.3f  sustained %.3f  of_fill %.3f  of_mix %s' % ('$w', j['ms_per_step'], r['kernel_avg_ms'], r['frac'], j['sustained']['frac'], r['frac_of_fill'], r['frac_of_copy_mix']))"; done | tee $OUT/bench_lines.txt
