#!/usr/bin/env python3
"""Every bench workload at 1x, 2x, 4x its BASELINE batch size (more sequences, same lengths): does the fraction of the roof hold when the
working set outgrows the caches?  (round 5: the two-pass one-hot did not, until its id scratch was sliced.)   size_sweep.py [WORKLOADS] [FACTORS]"""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bsq_bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
from bioseq_amd import capi, synth
lib = capi.load(); dev = torch.device("cuda:0"); torch.cuda.set_device(dev); stream = torch.cuda.current_stream()
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["cfg3", "cfg3b", "cfg3bcl", "cfg4f", "cfg4b", "cfg2", "cfg2sf", "cfg5", "cfg5aug"]
factors = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4]
for kv in os.environ.get("KNOBS", "").split(","):   # KNOBS=name=value,name=value
    if kv:
        capi.check(lib.bsq_tuning_set(kv.split("=")[0].encode(), int(kv.split("=")[1])))
        print("knob", kv, flush=True)
for w in names:
    row = []
    for f in factors:
        n = synth.CONFIGS[bench.WORKLOADS[w][0]]["n"] * f
        b = bench.Batch(w, lib, dev, stream, n=n)
        ok = b.check().get("ok")
        bench.ramp(b.step, stream)
        ms = bench.timed_loop(b.step, 30, 10, stream)
        row.append("%dx (%d seqs, %.2f GB out): %.1f us frac %.3f %s" % (f, n, b.out_bytes / 1e9, ms * 1e3, b.algo_bytes / (ms * 1e-3) / 8e12, "ok" if ok else "CHECK FAILED"))
        del b; torch.cuda.empty_cache()
    print("%-8s %s" % (w, " | ".join(row)), flush=True)
