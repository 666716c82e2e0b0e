#!/bin/bash
# full GPU suite + headline bench line (with e2e / sustained objects)
OUT=gpurun_out/${1:-r03b}; mkdir -p $OUT
( time timeout 3000 python -m pytest tests -m gpu -x -q ) > $OUT/gputest.txt 2>&1
tail -5 $OUT/gputest.txt
python3 bench.py > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err; echo "bench rc=$?"
python3 -c "
import json; j=json.load(open('$OUT/bench_cfg3.json'))
print('value', j['value'], 'ms/step', j['ms_per_step'], 'frac', j['roofline']['frac'], 'fill', j['roofline']['fill_yardsticks_gbps'], 'frac_of_fill', j['roofline']['frac_of_fill'])
print('sustained', j.get('sustained')); print('e2e', j.get('e2e'))"
for w in cfg2 cfg5; do python3 bench.py --workload $w --no-cpu-baseline > $OUT/bench_$w.json 2>> $OUT/bench_other.err; python3 -c "
import json; j=json.load(open('$OUT/bench_$w.json')); r=j['roofline']
print('$w', 'loop us %.2f' % (r['kernel_avg_ms']*1e3), 'frac %.3f' % r['frac'], 'mix yardstick', r['copy_mix_yardstick_gbps'], 'frac_of_mix', r['frac_of_copy_mix'], 'event floor us', r['event_pair_floor_ms']*1e3, 'sustained', j['sustained']['frac'])
print('e2e', j.get('e2e'))"; done
