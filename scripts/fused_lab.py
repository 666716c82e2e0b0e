#!/usr/bin/env python3
"""Round 3: where does the fused augment + tokenize launch (k_augment_tokens_fused) spend its time?  cfg5 batch."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioseq_amd import capi, synth
lib = capi.load(); dev = torch.device("cuda:0")
cfg = synth.CONFIGS["cfg5"]; B, P = cfg["n"], cfg["padlen"]
chars, offs = synth.synth_packed(cfg["seed"], B, cfg["lo"], cfg["hi"], cfg["letters"])
desc = capi.make_desc(cfg["key"], cfg["eos"], cfg["bos"], cfg["padchar"])
dch, dof = torch.from_numpy(chars).to(dev), torch.from_numpy(offs).to(dev)
out = torch.empty((B, P), dtype=torch.int8, device=dev)
seed = [0]
def loop_us(fn, n=60, warm=300, reps=4):
    for _ in range(warm): fn()
    res = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize(); res.append(a.elapsed_time(b) * 1e3 / n)
    return min(res), float(np.median(res))
def fused(frac, chain=1):
    def f():
        seed[0] += 1
        capi.check(lib.bsq_augment_tokenize_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), B, P, 1, 0, out.data_ptr(), chain, frac, ctypes.c_uint64(seed[0]), None))
    return f
def tokens(): capi.check(lib.bsq_tokenize_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), B, P, 1, 0, out.data_ptr(), None))
def aug(frac):
    def f():
        seed[0] += 1
        capi.check(lib.bsq_augment_device(dch.data_ptr(), dof.data_ptr(), B, 1, frac, ctypes.c_uint64(seed[0]), None))
    return f
for rnd in range(2):
    for name, fn in (("tokens only", tokens), ("augment only frac 0.5", aug(0.5)), ("augment only frac 1e-12", aug(1e-12)),
                     ("fused frac 0.5", fused(0.5)), ("fused frac 1e-12 (aug role exits at once)", fused(1e-12)), ("fused frac 1.0", fused(1.0)), ("fused frac 0.1", fused(0.1))):
        mn, med = loop_us(fn)
        print("  %-46s min %6.2f median %6.2f us" % (name, mn, med), flush=True)
