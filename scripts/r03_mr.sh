#!/bin/bash
# 2 ranks sharing the one GPU over gloo: the self-launching N > 1 path of bench.py incl. every gather form (never a measurement)
OUT=gpurun_out/r03c; mkdir -p $OUT
export BSQ_BENCH_BACKEND=gloo BSQ_BENCH_SHARE_GPU=1
for args in "--workload cfg1oh" "--workload cfg1oh --scaling strong --gather 2" "--workload cfg2 --scaling strong --gather 2" "--workload cfg3bcl --gather 1 --steps 5 --warmup 2"; do
  echo "== bench.py --gpus 2 $args"; timeout 900 python3 bench.py --gpus 2 --steps 5 --warmup 3 $args 2>$OUT/mr.err | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('rc ok; world', j['config']['rccl_world_size'], 'value', round(j['value'],3), 'gather', {k:(round(v['ms'],3)) for k,v in (j.get('gather') or {}).get('forms',{}).items()})" || tail -5 $OUT/mr.err
done 2>&1 | tee $OUT/multirank_selflaunch.txt
python -m pytest tests/test_sharding_ipc.py -m gpu -q 2>&1 | tail -2
