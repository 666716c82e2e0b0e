#!/usr/bin/env python3
"""(P,B) int8 token matrices (batch_tokenize's default layout): k_tokens_raw with the 256 x 64 tile (raw_mode 1) vs the
wide 1024 x 16 tile (raw_mode 4), interleaved, on the BASELINE batches and a few small ones.  Event-timed."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioseq_amd import capi, synth
lib = capi.load()
dev = torch.device("cuda:0")
SHAPES = [("cfg2", None), ("cfg4", None), ("cfg5", None), ("cfg1", None),
          ("AMINO20", (4096, 50, 510, 512)), ("AMINO20", (16384, 50, 510, 512)), ("AMINO20", (16384, 10, 60, 64)),
          ("DNA", (100000, 100, 250, 256)), ("AMINO20", (65000, 50, 1000, 1001))]
for name, shp in SHAPES:
    if shp is None:
        c = synth.CONFIGS[name]
        chars, offs = synth.synth_packed(c["seed"], c["n"], c["lo"], c["hi"], c["letters"])
        desc = capi.make_desc(c["key"], c["eos"], c["bos"], c["padchar"]); B, P = c["n"], c["padlen"]
    else:
        B, lo, hi, P = shp
        chars, offs = synth.synth_packed(7, B, lo, hi, synth.AA if name[0] == "A" else "ACGT")
        desc = capi.make_desc(name, 0, 0, 0)
    dch, dof = torch.from_numpy(chars).to(dev), torch.from_numpy(offs).to(dev)
    out = torch.empty(B * P, dtype=torch.uint8, device=dev)
    algo = int(offs[-1]) + 8 * (B + 1) + B * P
    def run(): capi.check(lib.bsq_tokenize_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), B, P, 0, 0, out.data_ptr(), None))
    for rnd in range(3):
        row = []
        for rm in (1, 4):
            capi.check(lib.bsq_tuning_set(b"raw_mode", rm))
            for _ in range(5): run()
            ts = []
            for _ in range(7):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(10): run()
                b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 10)
            ms = float(np.median(ts))
            row.append("raw_mode %d: %7.1f us %5.0f GB/s" % (rm, ms * 1e3, algo / ms / 1e6))
        print("%-8s B=%7d P=%5d | %s" % (name, B, P, " | ".join(row)), flush=True)
