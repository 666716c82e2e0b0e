#!/usr/bin/env python3
"""(P,B) token matrices of 2- / 4- / 8-byte elements (k_tokenize_tile): sequences per tile (knob tokenize_tb) x tile order,
on batch sizes whose rows are / are not 64-byte aligned."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioseq_amd import capi, synth
lib = capi.load()
dev = torch.device("cuda:0")
def timeit(fn):
    for _ in range(5): fn()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): fn()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 10)
    return float(np.median(ts))
buf = torch.empty(2**30, dtype=torch.uint8, device=dev)
for key, B, P in (("AMINO20", 65536, 1024), ("AMINO20", 65000, 1024), ("AMINO20", 65537, 1000), ("DNA", 250000, 256), ("DNA", 250001, 250), ("DNA", 1000, 256), ("AMINO20", 8192, 512)):
    chars, offs = synth.synth_packed(11, B, 30, P - 2, synth.AA if key[0] == "A" else "ACGT")
    dch, dof = torch.from_numpy(chars).to(dev), torch.from_numpy(offs).to(dev)
    desc = capi.make_desc(key, 0, 0, 0)
    for dc in "hil":
        dt = ctypes.c_int(0); capi.check(lib.bsq_dtype_from_destchar(dc.encode(), ctypes.byref(dt)))
        fn = lambda: capi.check(lib.bsq_tokenize_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), B, P, 0, dt, buf.data_ptr(), None))
        row = []
        for order in (0, 4):
            for tb in (64, 256):
                capi.check(lib.bsq_tuning_set(b"tile_order", order)); capi.check(lib.bsq_tuning_set(b"tokenize_tb", tb))
                row.append("o%d tb%d %.0f" % (order, tb, timeit(fn) * 1e3))
        print("%-7s B=%6d P=%4d %s | %s us" % (key, B, P, dc, " | ".join(row)), flush=True)
