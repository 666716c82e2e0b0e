#!/bin/bash
# Run ON the GPU box: tile order 0 / 1 / 2 for the tiled kernels (cfg4b: k_onehot_tile; cfg4f / cfg3: k_tokens_raw of the two-pass path)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for W in cfg4b cfg4f cfg3; do
  for o in 0 1 2; do
    OUT=$REPO/gpurun_out/order_${W}_$o; mkdir -p $OUT
    BSQ_TILE_ORDER=$o timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --full-line --workload $W --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2>/dev/null
    BSQ_TILE_ORDER=$o timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --full-line --workload $W --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
    python3 $REPO/scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
    echo "== $W order $o"; grep -E "k_tokens_raw|k_onehot_tile|k_expand" $OUT/summary.txt | grep -E "avg_ns|avg'" | cut -c1-260
  done
done
