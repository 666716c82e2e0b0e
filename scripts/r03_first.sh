#!/bin/bash
# round 3, first GPU call: the new reference-pinned tests, the self-launching N > 1 bench path, the headline line
OUT=gpurun_out/r03a; mkdir -p $OUT
python -m pytest tests/test_decode_device.py tests/test_facade.py tests/test_augment.py tests/test_every_device.py -m gpu -x -q 2>&1 | tail -15 > $OUT/new_tests.txt
export BSQ_BENCH_BACKEND=gloo BSQ_BENCH_SHARE_GPU=1
( timeout 600 python3 bench.py --gpus 2 --workload cfg1oh --steps 5 --warmup 3; echo "rc=$?" ) > $OUT/selflaunch_cfg1oh.txt 2> $OUT/selflaunch_cfg1oh.err
( timeout 600 python3 bench.py --gpus 2 --workload cfg1oh --steps 5 --warmup 3 --scaling strong --gather 2; echo "rc=$?" ) > $OUT/selflaunch_cfg1oh_strong.txt 2> $OUT/selflaunch_cfg1oh_strong.err
unset BSQ_BENCH_BACKEND BSQ_BENCH_SHARE_GPU
python3 bench.py --steps 20 --warmup 20 > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
for w in cfg2 cfg5 cfg4b cfg5aug; do python3 bench.py --workload $w --no-cpu-baseline > $OUT/bench_$w.json 2>> $OUT/bench_other.err; done
cat $OUT/new_tests.txt; tail -2 $OUT/selflaunch_cfg1oh.txt | cut -c1-300; tail -2 $OUT/selflaunch_cfg1oh_strong.txt | cut -c1-300
