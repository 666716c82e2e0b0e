#!/bin/bash
# Run ON a 1-GPU box: exercises bench.py's N > 1 code path (strong + weak scaling, every gather form) with two ranks
# that share device 0 over gloo.  Never a measurement -- the driver's 8-GPU runs use nccl = RCCL, one GPU per rank.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export BSQ_BENCH_BACKEND=gloo BSQ_BENCH_SHARE_GPU=1
for sc in weak strong; do
  timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
      "$REPO/bench.py" --gpus 2 --steps 5 --warmup 3 --workload ${1:-cfg2} --scaling $sc --gather 2 2>&1 | grep -E '^\{|Error|error' | tail -3
done
