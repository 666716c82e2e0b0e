#!/bin/bash
# Run ON the GPU box (through gpurun): kernel-trace stats + separate PMC passes for the bench command.
# Usage: scripts/profile_gpu.sh <tag> [bench args...]      outputs under gpurun_out/<tag>/
set -u
TAG=${1:-prof}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-sustained $*"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
echo "trace rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_pmc_write.json" 2> "$OUT/pmc_write.err"
echo "pmc write rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
echo "pmc fetch rc=$?"
find "$OUT" -name "*.csv" | head -20
python3 "$REPO/scripts/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
