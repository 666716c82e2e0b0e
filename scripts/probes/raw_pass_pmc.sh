cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for k in 0 1; do
  export BSQ_TOKENS_PB8_PAIR=$k
  O=$R/gpurun_out/rawpmc_$k; mkdir -p $O
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $R/scripts/probes/raw_pass_target.py > /dev/null 2> $O/err
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/sq1 -- python3 $R/scripts/probes/raw_pass_target.py > /dev/null 2>> $O/err
  timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/tcc -- python3 $R/scripts/probes/raw_pass_target.py > /dev/null 2>> $O/err
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/scripts/probes/raw_pass_target.py > /dev/null 2>> $O/err
  timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $O/sq2 -- python3 $R/scripts/probes/raw_pass_target.py > /dev/null 2>> $O/err
  echo "=== pair knob $k (0 = paired tail, 1 = unpaired)"
  python3 - $O <<'PY'
import csv, glob, os, sys
import numpy as np
from collections import defaultdict
O = sys.argv[1]
for f in glob.glob(os.path.join(O, "t", "**", "*kernel_trace.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "k_tokens_pb8" in r["Kernel_Name"]]
    d = np.array([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]) / 1e3
    print("  raw pass: resident (launches 5..40) %.2f us, cycling over 6 batches (launches 46..100) %.2f us" % (d[5:40].mean(), d[46:100].mean()))
for sub in ("sq1", "sq2", "tcc", "fetch"):
    for f in glob.glob(os.path.join(O, sub, "**", "*counter_collection.csv"), recursive=True):
        d = defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_tokens_pb8" in r["Kernel_Name"]:
                d[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in sorted(d.items()):
            v = np.array(v)
            print("   %-22s resident %.4g   cycling %.4g" % (c, v[5:40].mean(), v[46:100].mean()))
PY
  rm -rf $O/t $O/sq1 $O/sq2 $O/tcc $O/fetch
done
