#!/usr/bin/env python3
"""One knob, several values, interleaved in one process: bench.py's loop / sustained numbers of workloads (each value checked against the
reference's folds first).      [COLD=1] knob_lab.py KNOB v1,v2,... WORKLOAD [WORKLOAD ...]"""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bsq_bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
import torch
from bioseq_amd import capi
lib = capi.load()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
stream = torch.cuda.current_stream()
knob, vals = sys.argv[1].encode(), [int(v) for v in sys.argv[2].split(",")]
for w in sys.argv[3:]:
    b = bench.Batch(w, lib, dev, stream)
    for v in vals:
        capi.check(lib.bsq_tuning_set(knob, v))
        print(w, "%s=%d check:" % (knob.decode(), v), b.check().get("ok"), flush=True)
    for rnd in range(3):
        row = []
        for v in vals:
            capi.check(lib.bsq_tuning_set(knob, v))
            bench.ramp(b.step, stream)
            loop_ms = bench.timed_loop(b.step, 100, 30, stream)
            sus = bench.sustained_loop(b, 0.4, 100, loop_ms, stream)
            row.append("%d: loop %.1f us, sustained %.1f us (frac %.3f)" % (v, loop_ms * 1e3, sus["kernel_avg_ms"] * 1e3, sus["frac"]))
        print("  %s round %d  %s" % (w, rnd, " | ".join(row)), flush=True)
    if os.environ.get("COLD"):   # the cold regime of bench.py (inputs and outputs cycling over > 512 MiB), per value, interleaved
        for rnd in range(2):
            row = []
            for v in vals:
                capi.check(lib.bsq_tuning_set(knob, v))
                c = bench.cold_regime(b, 200, 0.3, stream)
                row.append("%d: cold %.2f us (frac %.3f), sustained %.2f us" % (v, c["ms_per_step"] * 1e3, c["frac"], c["sustained_ms_per_step"] * 1e3))
            print("  %s cold round %d  %s" % (w, rnd, " | ".join(row)), flush=True)
    capi.check(lib.bsq_tuning_set(knob, 0))
    del b
    torch.cuda.empty_cache()
