#!/usr/bin/env python3
"""Where does the cold-input regime of the (B,P) int8 token kernel lose against its stream yardstick?  The ablation ladder of
k_tokens_bp8 (knob tokens8_abl; needs the DIAGNOSTIC build, scripts/build_labs.sh -- its outputs are wrong on purpose) on a resident
batch and cycling over > 512 MiB of distinct batches:
  0 the kernel | 1 no alphabet lookup | 5 character loads at addresses that do not depend on the offsets | 2 no character loads |
  3 no offsets loads either | 4 stores only            (tokens8_fast = 1: the ladder exists for the plain kernel, not the fast form)"""
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
for w in sys.argv[1:] or ["cfg2", "cfg5"]:
    b = bench.Batch(w, lib, dev, stream)
    b.check()
    for rnd in range(2):
        for fast, abl in ((0, 0), (1, 0), (1, 1), (1, 5), (1, 2), (1, 3), (1, 4)):
            capi.check(lib.bsq_tuning_set(b"tokens8_fast", fast))
            capi.check(lib.bsq_tuning_set(b"tokens8_abl", abl))
            bench.ramp(b.step, stream)
            loop_ms = bench.timed_loop(b.step, 300, 50, stream)
            cold = bench.cold_regime(b, 300, 0.3, stream, verify=False)
            print("%s round %d  %-22s resident %.2f us   cold %.2f us   (copy-mix cold %.2f us)" % (
                w, rnd, "fast kernel" if not fast else "plain kernel, abl %d" % abl, loop_ms * 1e3, cold["sustained_ms_per_step"] * 1e3,
                cold.get("copy_mix_ms", 0) * 1e3), flush=True)
    capi.check(lib.bsq_tuning_set(b"tokens8_fast", 0))
    capi.check(lib.bsq_tuning_set(b"tokens8_abl", 0))
    del b
    torch.cuda.empty_cache()
