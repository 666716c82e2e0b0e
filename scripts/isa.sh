#!/bin/bash
# Device assembly + per-kernel register / LDS use of one .hip source (cross-compiles here, no GPU needed).
# Usage: scripts/isa.sh bsq_tokens8.hip [extra hipcc flags]   ->  /tmp/isa/<name>.s  + a resource table on stdout
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
SRC=$1; shift
mkdir -p /tmp/isa
OUT=/tmp/isa/$(basename "$SRC" .hip).s
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I"$REPO/include" -I"$REPO/bioseq_amd/csrc" -x hip -mllvm -amdgpu-kernarg-preload-count=14 \
      --cuda-device-only -S "$@" -o "$OUT" "$REPO/bioseq_amd/csrc/$SRC"
python3 - "$OUT" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"- \.agpr_count:.*?\.name:\s+(\S+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", txt, re.S):
    blk = m.group(0)
    lds = re.search(r"\.group_segment_fixed_size:\s+(\d+)", blk)
    print("%-110s vgpr %3s sgpr %3s spill %s lds %s" % (m.group(1)[:110], m.group(3), m.group(2), m.group(4), lds.group(1) if lds else "?"))
PY
