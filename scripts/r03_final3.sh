#!/bin/bash
# evidence of the workloads whose kernels changed with k_tokens_pb8_fast (cfg3, cfg4f, cfg2sf): bench lines, rocprofv3 stats + WRITE/FETCH passes, SQ/TCC counters
OUT=gpurun_out/r03g; mkdir -p $OUT
python3 bench.py > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
for w in cfg4f cfg2sf; do python3 bench.py --workload $w --no-cpu-baseline > $OUT/bench_$w.json 2>> $OUT/bench.err; done
for w in cfg3 cfg4f cfg2sf; do python3 -c "
import json; j=json.load(open('$OUT/bench_$w.json')); r=j['roofline']; e=j.get('e2e') or {}
print('%-8s ms/step %.4f  loop %.4f ms  frac %.3f  sustained %.4f ms frac %.3f  of_fill %.3f  kernel %s' % ('$w', j['ms_per_step'], r['kernel_avg_ms'], r['frac'], j['sustained']['kernel_avg_ms'], j['sustained']['frac'], r['frac_of_fill'], r['kernel']))"; done | tee $OUT/bench_lines.txt
WORKLOADS="cfg3 cfg4f cfg2sf" bash scripts/evidence_all.sh r03 > $OUT/evidence.log 2>&1
bash scripts/r03_pmc.sh r03_sq_cfg2sf --workload cfg2sf > /dev/null 2>&1
echo done
