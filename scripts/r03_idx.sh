#!/bin/bash
OUT=gpurun_out/r03a; mkdir -p $OUT
timeout 900 python -m pytest tests/test_index_batches.py tests/test_bcl_and_loaders.py tests/test_flatfile.py tests/test_augment.py -m gpu -x -q 2>&1 | tail -25 | tee $OUT/idx_tests.txt
