cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/staged_tl; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python3 $R/scripts/probes/staged_timeline_target.py > /dev/null 2> $O/trace.err
python3 - $O <<'PY' | tee $O/summary.txt
import csv, glob, os, sys
ks = []
for f in glob.glob(os.path.join(sys.argv[1], "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], "K"))
for f in glob.glob(os.path.join(sys.argv[1], "trace", "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", r.get("Kind", "")) , "C"))
ks.sort()
# the LAST call: everything after the last gap of more than 20 ms
cut = 0
for i in range(1, len(ks)):
    if ks[i][0] - ks[i - 1][1] > 20_000_000:
        cut = i
last = ks[cut:]
t0 = last[0][0]
prev_k_end = None
for s, e, name, kind in last:
    gap = ""
    if kind == "K":
        if prev_k_end is not None:
            gap = "  (%.0f us after the previous kernel ended)" % ((s - prev_k_end) / 1e3)
        prev_k_end = e
    print("%9.1f us  +%8.1f us  %-62s%s" % ((s - t0) / 1e3, (e - s) / 1e3, name, gap))
PY
rm -rf $O/trace
