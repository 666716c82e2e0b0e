#!/bin/bash
# Run ON the GPU box: ab/old.so vs ab/new.so, one forced one-hot path ($1) over the shapes given as further arguments.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
P=$1; shift
for rep in 1 2; do
  for v in old new; do
    cp ab/$v.so bioseq_amd/libbsq_hip.so
    for si in "$@"; do echo "$v shape $si: $(python3 scripts/ab_knob.py $si $P onehot_path $P 2>&1 | grep -v amdgpu | tail -1)"; done
  done
done
cp ab/new.so bioseq_amd/libbsq_hip.so
