#!/usr/bin/env python3
"""What the BLOSUM62 augmentation kernel's time is made of: cfg5 batch, augment_frac 0 (table copy + spans only) .. 1,
chain_len 1 / 2.  Event-timed, in place on the same batch."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioseq_amd import capi, synth
lib = capi.load()
dev = torch.device("cuda:0")
c = synth.CONFIGS["cfg5"]
chars, offs = synth.synth_packed(c["seed"], c["n"], c["lo"], c["hi"], c["letters"])
dch, dof = torch.from_numpy(chars).to(dev), torch.from_numpy(offs).to(dev)
B = c["n"]
def timeit(fn):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): fn()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 10)
    return float(np.median(ts))
for mode, spw in ((0, 64), (1, 0)):  # (the experimental build swept 64 / 32 / 16 / 8 sequences per wave: profiles/r02/augment_lab2.txt)
    capi.check(lib.bsq_tuning_set(b"augment_mode", mode))
    for chain in (1, 2):
        row = []
        for frac in (0.0, 0.1, 0.5, 1.0):
            seed = [0]
            def run():
                seed[0] += 1
                capi.check(lib.bsq_augment_device(dch.data_ptr(), dof.data_ptr(), B, chain, frac, ctypes.c_uint64(seed[0]), None))
            row.append("frac %.1f %.1f us" % (frac, timeit(run) * 1e3))
        print("augment_mode %d spw %2d chain_len %d | %s" % (mode, spw, chain, " | ".join(row)), flush=True)
