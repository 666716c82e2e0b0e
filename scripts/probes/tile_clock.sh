# Is the tiled one-hot kernel's box-to-box variance (25 %: profiles/r06/dispatch_check.txt) the CLOCK or the MEMORY side?  Per box: durations
# from a kernel trace, GRBM_GUI_ACTIVE / SQ cycles per launch from a counter pass (cycles / duration = the clock the kernel ran at), the
# memory side's request and stall counts from another, and rocm-smi's clocks and power before and after.  One box per call: compare across calls.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/tileclk; mkdir -p $O
T=$R/scripts/probes/tile_clock_target.py
echo "=== $(date -u +%FT%TZ) $(hostname)"
rocm-smi --showclocks --showpower --showperflevel 2>/dev/null | grep -v "^=\|^$" | head -20
python3 $T
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|fclk\|power" | head -8
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $T > /dev/null 2> $O/err
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $O/c1 -- python3 $T > /dev/null 2>> $O/err
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $O/c2 -- python3 $T > /dev/null 2>> $O/err
timeout 300 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/c3 -- python3 $T > /dev/null 2>> $O/err
timeout 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $O/c4 -- python3 $T > /dev/null 2>> $O/err
python3 - $O <<'PY'
import csv, glob, os, sys
import numpy as np
from collections import defaultdict
O = sys.argv[1]
KERN = ("k_onehot_tile", "k_tokens_pb8", "k_expand", "erfinv")
def key_of(name):
    for k in KERN:
        if k in name:
            return k
    return None
def grid(r):
    return r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
dur = defaultdict(list)
for f in glob.glob(os.path.join(O, "t", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = key_of(r["Kernel_Name"])
        if k:
            dur[(k, grid(r))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("kernel trace (no counters): mean / min / max us per launch, by kernel and grid")
for (k, g), v in sorted(dur.items()):
    v = np.array(v)
    print("  %-16s grid %-10s n %3d  %8.1f / %8.1f / %8.1f" % (k, g, len(v), v.mean(), v.min(), v.max()))
for sub in ("c1", "c2", "c3", "c4"):
    for f in glob.glob(os.path.join(O, sub, "**", "*counter_collection.csv"), recursive=True):
        d = defaultdict(lambda: defaultdict(list))
        t = defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = key_of(r["Kernel_Name"])
            if not k:
                continue
            d[(k, grid(r))][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r and r.get("End_Timestamp"):
                t[(k, grid(r))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for kg, cs in sorted(d.items()):
            tt = np.array(t[kg]).mean() if t[kg] else float("nan")
            line = "  %-16s grid %-10s (%.1f us under counters) " % (kg[0], kg[1], tt)
            line += "  ".join("%s %.4g" % (c, np.array(v).mean()) for c, v in sorted(cs.items()))
            if "GRBM_GUI_ACTIVE" in cs and tt == tt:
                line += "   => GRBM_GUI_ACTIVE / duration = %.0f MHz" % (np.array(cs["GRBM_GUI_ACTIVE"]).mean() / tt)
            print(line)
PY
rm -rf $O
