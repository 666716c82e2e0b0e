#!/usr/bin/env python3
"""Nibble id scratch (knob raw_nibbles: 0 nibbles where they apply, 1 bytes) x the expansion's pacing load (expand_gate: 0 automatic, 1 never,
2 always) x occupancy pad, interleaved on bench workloads; every arm checked against the reference's folds first.
    nib_lab.py WORKLOAD [WORKLOAD ...]"""
import importlib.util, itertools, os, sys
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
pads = [int(x) for x in os.environ.get("PADS", "0").split(",")]
arms = list(itertools.product((1, 0), (0, 1, 2), pads))
def setk(nib, gate, pad):
    capi.check(lib.bsq_tuning_set(b"raw_nibbles", nib)); capi.check(lib.bsq_tuning_set(b"expand_gate", gate)); capi.check(lib.bsq_tuning_set(b"expand_pad", pad))
for w in sys.argv[1:]:
    b = bench.Batch(w, lib, dev, stream)
    for a in arms:
        setk(*a)
        assert b.check().get("ok"), (w, a)
    for rnd in range(2):
        row = []
        for a in arms:
            setk(*a)
            bench.ramp(b.step, stream)
            loop_ms = bench.timed_loop(b.step, 60, 20, stream)
            row.append("%s gate%d pad%d: %.1f us (%.3f)" % ("bytes" if a[0] else "nibbles", a[1], a[2], loop_ms * 1e3, b.algo_bytes / (loop_ms * 1e-3) / 8e12))
        print("  %s round %d  %s" % (w, rnd, " | ".join(row)), flush=True)
    setk(0, 0, 0)
    del b
    torch.cuda.empty_cache()
