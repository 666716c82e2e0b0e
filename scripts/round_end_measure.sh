#!/bin/bash
# Run ON the GPU box (through gpurun): the round-end measurement pass -- GPU suite, rocprofv3 profile of the default bench,
# bench lines of every workload, shape sweep.  Results land in gpurun_out/; copy what is to be judged into profiles/.
python -m pytest tests -x -q -m gpu 2>&1 | tail -2
bash scripts/profile_gpu.sh r01_cfg3_v6 > gpurun_out/prof_v6.log 2>&1
python bench.py > gpurun_out/bench_final_cfg3.json 2> gpurun_out/bench_final_cfg3.err
for w in cfg2 cfg4f cfg4b cfg5 cfg5aug cfg3bcl; do python bench.py --workload $w > gpurun_out/bench_final_$w.json 2>/dev/null; done
python scripts/sweep_shapes.py > gpurun_out/sweep_shapes5.txt 2>&1
tail -1 gpurun_out/bench_final_cfg3.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['gb_per_s_written'], d['ms_per_step'], d['roofline'], d.get('cpu_baseline'))"
tail -30 gpurun_out/prof_v6.log
