#!/bin/bash
# k_tokens_pb8_fast for every element type: parity, then batch_tokenize over dtypes x layouts on the cfg2 batch with and without it
OUT=gpurun_out/r03pbw; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_tokens_seqfirst.py -m gpu -x -q 2>&1 | tail -4 | tee $OUT/tests.txt
for v in 1 0; do echo "== tokens_pb8=$v"; BSQ_TOKENS_PB8=$v python3 scripts/tokens_dtypes.py 2>/dev/null; done | tee $OUT/tokens_dtypes.txt
