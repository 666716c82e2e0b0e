#!/bin/bash
# Run ON the GPU box: expand_gate 1 (never) vs 2 (always) on the shapes of sweep_shapes_list.py through the two-pass path (onehot_path 2)
cd ${GRAFT_REPO_ROOT:-.}
for si in "$@"; do python3 scripts/ab_knob.py $si 2 expand_gate 1 2 2>&1 | grep -v amdgpu | awk 'NR==1 || /round 2/'; done
