#!/usr/bin/env python3
"""batch_tokenize throughput over dtypes and layouts on the cfg2 batch (65 536 x 1024): event-timed, input resident."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioseq_amd import capi, synth
lib = capi.load()
dev = torch.device("cuda:0")
c = synth.CONFIGS["cfg2"]
chars, offs = synth.synth_packed(c["seed"], c["n"], c["lo"], c["hi"], c["letters"])
dch, dof = torch.from_numpy(chars).to(dev), torch.from_numpy(offs).to(dev)
desc = capi.make_desc(c["key"], c["eos"], c["bos"], c["padchar"])
B, P = c["n"], c["padlen"]
for dc in "bhifdl":
    dt = ctypes.c_int(0); capi.check(lib.bsq_dtype_from_destchar(dc.encode(), ctypes.byref(dt)))
    sz = lib.bsq_dtype_size(dt)
    out = torch.empty(B * P * sz, dtype=torch.uint8, device=dev)
    algo = int(offs[-1]) + 8 * (B + 1) + B * P * sz
    res = []
    for bf in (1, 0):
        def run(): capi.check(lib.bsq_tokenize_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), B, P, bf, dt, out.data_ptr(), None))
        for _ in range(10): run()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): run()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 50
        res.append("%s %7.1f us %5.0f GB/s" % ("(B,P)" if bf else "(P,B)", ms * 1e3, algo / ms / 1e6))
    print("destchar %s (%d B): %s" % (dc, sz, " | ".join(res)), flush=True)
