#!/bin/bash
# Run ON the GPU box: ab/old.so vs ab/new.so on the tiled one-hot kernel (forced path 1) over a few shapes, interleaved.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
for rep in 1 2; do
  for v in old new; do
    cp ab/$v.so bioseq_amd/libbsq_hip.so
    for si in 13 12 2 40 41 38 32; do echo "$v shape $si: $(python3 scripts/ab_knob.py $si 1 onehot_path 1 2>&1 | grep -v amdgpu | tail -1)"; done
  done
done
cp ab/new.so bioseq_amd/libbsq_hip.so
