#!/bin/bash
# Run ON the GPU box: rocprofv3 kernel stats + separate WRITE_SIZE / FETCH_SIZE passes of bench.py for EVERY workload.
# Usage: scripts/evidence_all.sh <round-tag, e.g. r02>   -> gpurun_out/<tag>_<workload>/{summary.txt,summary.json,...}
# Back in the build container: python scripts/make_traffic.py <tag>  copies the summaries into profiles/<tag>/ and
# rebuilds profiles/traffic.json (what bench.py reports as roofline.traffic).
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for w in ${WORKLOADS:-cfg3 cfg2 cfg5 cfg4f cfg4b cfg5aug cfg3bcl}; do
  bash "$REPO/scripts/profile_gpu.sh" ${TAG}_$w --workload $w > "$REPO/gpurun_out/${TAG}_$w.log" 2>&1
  echo "$w: $(grep -c . "$REPO/gpurun_out/${TAG}_$w/summary.txt") summary lines"
done
