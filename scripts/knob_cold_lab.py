#!/usr/bin/env python3
"""One knob, several values, interleaved in one process: bench.py's loop / sustained / COLD-regime numbers of a workload.
    knob_cold_lab.py WORKLOAD KNOB v1 v2 ...        e.g.  knob_cold_lab.py cfg2 tokens8_ahead 0 32 64 128 256"""
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
w, knob, vals = sys.argv[1], sys.argv[2].encode(), [int(v) for v in sys.argv[3:]]
b = bench.Batch(w, lib, dev, stream)
b.check()
for rnd in range(3):
    for v in vals:
        capi.check(lib.bsq_tuning_set(knob, v))
        bench.ramp(b.step, stream)
        loop_ms = bench.timed_loop(b.step, 200, 50, stream)
        sus = bench.sustained_loop(b, 0.3, 200, loop_ms, stream)
        cold = bench.cold_regime(b, 200, 0.4, stream)
        print("round %d  %s=%-4d loop %.2f us  sustained %.2f us (frac %.3f)  cold %.2f us (frac %.3f; copy-mix %s)" % (
            rnd, knob.decode(), v, loop_ms * 1e3, sus["kernel_avg_ms"] * 1e3, sus["frac"], cold["sustained_ms_per_step"] * 1e3,
            cold["frac_sustained"], ("%.2f us" % (cold["copy_mix_ms"] * 1e3)) if "copy_mix_ms" in cold else "-"), flush=True)
capi.check(lib.bsq_tuning_set(knob, 0))
