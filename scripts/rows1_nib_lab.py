#!/usr/bin/env python3
"""int8 one-hots with tiny rows (cfg4b: 7-byte rows, 1M reads): the tiled launch against the two-pass forms -- byte ids + k_expand_rows1,
NIBBLE ids + k_expand_rows1<nibbles> -- interleaved in one process, loop / sustained and the COLD regime of bench.py; each form checked
against the reference's folds first.      rows1_nib_lab.py [WORKLOAD ...]   (default cfg4b)"""
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
FORMS = [("tiled", {"onehot_path": 1}), ("two-pass bytes", {"onehot_path": 2, "raw_nibbles": 1}), ("two-pass nibbles", {"onehot_path": 2, "raw_nibbles": 2}),
         ("automatic", {})]
KNOBS = ("onehot_path", "raw_nibbles")


def apply(form):
    for k in KNOBS:
        capi.check(lib.bsq_tuning_set(k.encode(), form.get(k, 0)))


for w in sys.argv[1:] or ["cfg4b"]:
    b = bench.Batch(w, lib, dev, stream)
    for name, form in FORMS:
        apply(form)
        print(w, name, "check:", b.check().get("ok"), flush=True)
    for rnd in range(3):
        for name, form in FORMS:
            apply(form)
            bench.ramp(b.step, stream)
            loop_ms = bench.timed_loop(b.step, 100, 30, stream)
            sus = bench.sustained_loop(b, 0.4, 100, loop_ms, stream)
            print("  %s round %d  %-18s loop %.1f us, sustained %.1f us (frac %.3f)" % (w, rnd, name, loop_ms * 1e3, sus["kernel_avg_ms"] * 1e3, sus["frac"]), flush=True)
    for rnd in range(2):
        for name, form in FORMS:
            apply(form)
            c = bench.cold_regime(b, 200, 0.3, stream)
            print("  %s cold round %d  %-18s %.2f us (frac %.3f), sustained %.2f us" % (w, rnd, name, c["ms_per_step"] * 1e3, c["frac"], c["sustained_ms_per_step"] * 1e3), flush=True)
    apply({})
    del b
    torch.cuda.empty_cache()
