# The tiled one-hot kernel's time follows the result's ALLOCATION (tile_placement.py).  Is it address translation?  UTCL1 (per-CU TLB) hit / miss
# counts, UTCL2 busy cycles and the memory side's write stalls per launch, by allocation, in ONE process that has a fast and a slow allocation.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/tileplc; mkdir -p $O
T=$R/scripts/probes/tile_placement_target.py
python3 $T 2>&1 | grep -v amdgpu.ids
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $T > $O/t.out 2> $O/err
timeout 300 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum --output-format csv -d $O/c1 -- python3 $T > $O/c1.out 2>> $O/err
timeout 300 rocprofv3 --pmc GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCC_EA0_WRREQ_STALL_sum --output-format csv -d $O/c2 -- python3 $T > $O/c2.out 2>> $O/err
timeout 300 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS TCP_UTCL1_SERIALIZATION_STALL TCP_UTCL1_THRASHING_STALL --output-format csv -d $O/c3 -- python3 $T > $O/c3.out 2>> $O/err
grep -i "error\|invalid\|not found\|unsupported" $O/err | sort | uniq -c | head -5
python3 - $O <<'PY'
import csv, glob, os, sys
import numpy as np
from collections import defaultdict
O = sys.argv[1]
LABELS = ["first", "filler3", "filler64", "filler513", "again0"]
N = 10
def show(sub):
    for line in open(os.path.join(O, sub + ".out")):
        if line.startswith("allocation"):
            print("    (this pass's own events) " + line.strip())
for f in glob.glob(os.path.join(O, "t", "**", "*kernel_trace.csv"), recursive=True):
    rows = sorted((r for r in csv.DictReader(open(f)) if "k_onehot_tile" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
    d = np.array([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]) / 1e3
    print("kernel trace: k_onehot_tile us per launch by allocation: " + "  ".join("%s %.1f" % (l, d[i * N + 2:(i + 1) * N].mean()) for i, l in enumerate(LABELS)))
    show("t")
for sub in ("c1", "c2", "c3"):
    for f in glob.glob(os.path.join(O, sub, "**", "*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "k_onehot_tile" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        by = defaultdict(list)
        for r in rows:
            by[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("counters per launch of k_onehot_tile by allocation (%s):" % sub)
        show(sub)
        for c, v in sorted(by.items()):
            v = np.array(v)
            print("  %-44s " % c + "  ".join("%s %.4g" % (l, v[i * N + 2:(i + 1) * N].mean()) for i, l in enumerate(LABELS)))
PY
rm -rf $O
