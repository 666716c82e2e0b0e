#!/bin/bash
# refresh of the closing evidence after the augmentation kernel's home-lane first round: GPU suite, bench lines of every workload, cfg5aug rocprof + counters
OUT=gpurun_out/r03y; mkdir -p $OUT
( time timeout 3000 python -m pytest tests -m gpu -q ) > $OUT/gputest.txt 2>&1
grep -E "passed|failed|error" $OUT/gputest.txt | tail -2
python3 bench.py > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
for w in cfg2 cfg5 cfg4f cfg4b cfg5aug cfg2sf cfg3bcl; do python3 bench.py --workload $w --no-cpu-baseline > $OUT/bench_$w.json 2>> $OUT/bench.err; done
for w in cfg3 cfg2 cfg5 cfg4f cfg4b cfg5aug cfg2sf cfg3bcl; do python3 -c "
import json; j=json.load(open('$OUT/bench_$w.json')); r=j['roofline']; e=j.get('e2e') or {}
print('%-8s ms/step %.4f  loop %.4f ms  frac %.3f  sustained %.4f ms frac %.3f  of_fill %.3f  of_mix %s  e2e list->dev %s ms (auto threads), pipelined %s, list->numpy %s' % ('$w', j['ms_per_step'], r['kernel_avg_ms'], r['frac'], j['sustained']['kernel_avg_ms'], j['sustained']['frac'], r['frac_of_fill'], r['frac_of_copy_mix'] and round(r['frac_of_copy_mix'],3), e.get('list_to_device_sync_default_nthreads_ms') and round(e['list_to_device_sync_default_nthreads_ms'],2), e.get('list_to_device_pipelined20_ms') and round(e['list_to_device_pipelined20_ms'],2), e.get('list_to_numpy_ms') and round(e['list_to_numpy_ms'],1)))"; done | tee $OUT/bench_lines.txt
WORKLOADS="cfg5aug" bash scripts/evidence_all.sh r03 > $OUT/evidence.log 2>&1
bash scripts/r03_pmc.sh r03_sq_cfg5aug --workload cfg5aug > /dev/null 2>&1
echo done
