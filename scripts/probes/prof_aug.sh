cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/aug_prof; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/scripts/probes/aug_profile_target.py > /dev/null 2> $O/trace.err
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/sq1 -- python3 $R/scripts/probes/aug_profile_target.py > /dev/null 2> $O/sq1.err
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM --output-format csv -d $O/sq2 -- python3 $R/scripts/probes/aug_profile_target.py > /dev/null 2> $O/sq2.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/scripts/probes/aug_profile_target.py > /dev/null 2> $O/fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/scripts/probes/aug_profile_target.py > /dev/null 2> $O/write.err
python3 - $O <<'PY' > $O/summary.txt
import csv, glob, os, sys
from collections import defaultdict
for f in glob.glob(os.path.join(sys.argv[1], "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print("%-90s calls %4s avg %10.2f us  %5s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
for sub in ("sq1", "sq2", "fetch", "write"):
    for f in glob.glob(os.path.join(sys.argv[1], sub, "**", "*counter_collection.csv"), recursive=True):
        d = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            d[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in d.items():
            if "k_augment" in k:
                print(k[:100])
                for c, v in sorted(cs.items()):
                    print("   %-24s avg %.4g (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
rm -rf $O/trace $O/sq1 $O/sq2 $O/fetch $O/write
cat $O/summary.txt
