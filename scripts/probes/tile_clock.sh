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
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/c3 -- python3 $T > /dev/null 2>> $O/err
timeout 300 rocprofv3 --pmc TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/c3b -- python3 $T > /dev/null 2>> $O/err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c4 -- python3 $T > /dev/null 2>> $O/err
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c5 -- python3 $T > /dev/null 2>> $O/err
grep -i "error\|invalid\|not found\|unsupported" $O/err | sort | uniq -c | head -5
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
class Shape:
    """the target runs shape 1 then shape 2, 60 one-hot calls each: the ordinal of a launch of the tiled kernel / the raw pass names its shape"""
    def __init__(self):
        self.n = defaultdict(int)
    def __call__(self, k, r, per_counter=1):
        if k not in ("k_onehot_tile", "k_tokens_pb8"):
            return grid(r)
        i = self.n[(k, r.get("Counter_Name", ""))]
        self.n[(k, r.get("Counter_Name", ""))] += 1
        return "dna4_i16" if i < 60 else "dna5_i8"
dur = defaultdict(list)
for f in glob.glob(os.path.join(O, "t", "**", "*kernel_trace.csv"), recursive=True):
    sh = Shape()
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    for r in rows:
        k = key_of(r["Kernel_Name"])
        if k:
            dur[(k, sh(k, r))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("kernel trace (no counters): mean / min / max us per launch, by kernel and grid")
for (k, g), v in sorted(dur.items()):
    v = np.array(v)
    print("  %-16s grid %-10s n %3d  %8.1f / %8.1f / %8.1f" % (k, g, len(v), v.mean(), v.min(), v.max()))
for (k, g), v in sorted(dur.items()):
    if k == "k_onehot_tile":
        print("  %s %s, launch by launch: %s" % (k, g, " ".join("%.0f" % x for x in v)))
for sub in ("c1", "c2", "c3", "c3b", "c4", "c5"):
    for f in glob.glob(os.path.join(O, sub, "**", "*counter_collection.csv"), recursive=True):
        d = defaultdict(lambda: defaultdict(list))
        t = defaultdict(list)
        sh = Shape()
        rows = list(csv.DictReader(open(f)))
        if rows and "Dispatch_Id" in rows[0]:
            rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        for r in rows:
            k = key_of(r["Kernel_Name"])
            if not k:
                continue
            g = sh(k, r)
            d[(k, g)][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r and r.get("End_Timestamp"):
                t[(k, g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for kg, cs in sorted(d.items()):
            tt = np.array(t[kg]).mean() if t[kg] else float("nan")
            line = "  %-16s grid %-10s (%.1f us under counters) " % (kg[0], kg[1], tt)
            line += "  ".join("%s %.4g" % (c, np.array(v).mean()) for c, v in sorted(cs.items()))
            if "GRBM_GUI_ACTIVE" in cs and tt == tt:
                line += "   => GRBM_GUI_ACTIVE / 8 XCDs / duration = %.0f MHz" % (np.array(cs["GRBM_GUI_ACTIVE"]).mean() / tt / 8)
            print(line)
PY
rm -rf $O
