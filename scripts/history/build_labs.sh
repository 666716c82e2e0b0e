#!/bin/bash
# Diagnostic build of libbsq_hip.so: -DBSQ_LABS compiles the experiment kernels that lost their measurement
# (k_expand_small, k_tokens_raw2, the claim / div64 / four-chunk variants, the round-1 k_augment) and the ablation
# variants whose output is wrong on purpose, and makes their knobs settable.  The lab scripts under scripts/ that use
# those knobs need it.  NEVER ship this build: run `python bioseq_amd/build.py` afterwards to restore the product library
# (the flag change forces a full rebuild either way).
cd "$(dirname "$0")/.." && BSQ_EXTRA_HIPCC_FLAGS="-DBSQ_LABS" python3 bioseq_amd/build.py "$@"
