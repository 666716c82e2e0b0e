#!/usr/bin/env python3
"""The expansion's occupancy pad (unused dynamic LDS), finely, on bench workloads: byte ids and nibble ids.  PADS=... NIBS=0,1 pad_fine_lab.py W..."""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bsq_bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
from bioseq_amd import capi
lib = capi.load(); dev = torch.device("cuda:0"); torch.cuda.set_device(dev); stream = torch.cuda.current_stream()
pads = [int(x) for x in os.environ["PADS"].split(",")]
nibs = [int(x) for x in os.environ.get("NIBS", "1").split(",")]
for w in sys.argv[1:]:
    b = bench.Batch(w, lib, dev, stream)
    for rnd in range(2):
        for nib in nibs:
            row = []
            for pd in pads:
                capi.check(lib.bsq_tuning_set(b"raw_nibbles", nib)); capi.check(lib.bsq_tuning_set(b"expand_pad", pd))
                if rnd == 0: assert b.check().get("ok"), (w, nib, pd)
                bench.ramp(b.step, stream)
                ms = bench.timed_loop(b.step, 60, 20, stream)
                row.append("%d: %.1f" % (pd, ms * 1e3))
            print("  %s round %d %s  %s" % (w, rnd, "bytes  " if nib else "nibbles", " | ".join(row)), flush=True)
    capi.check(lib.bsq_tuning_set(b"raw_nibbles", 0)); capi.check(lib.bsq_tuning_set(b"expand_pad", 0))
    del b; torch.cuda.empty_cache()
