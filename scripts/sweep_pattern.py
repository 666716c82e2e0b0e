#!/usr/bin/env python3
"""Which WRITE PATTERNS reach the HBM write roof?  (diagnostic, cfg3-shaped output matrix)"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioseq_amd import capi
lib = capi.load()
dev = torch.device("cuda:0")
rows = 1024
buf = torch.empty(rows * 5120 * 1100 + 4096, dtype=torch.uint8, device=dev)

def timeit(fn, n=5, reps=4):
    ts = []
    for _ in range(reps):
        fn(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / n)
    return float(np.median(ts))

print("pattern: pitch seg rpw order interleave(2=rowwise) nt -> ms, GB/s")
for mult in (1024, 1028, 1032, 1036, 1052):
    for seg in (5120, 1024, 4096):
        pitch = 5120 * mult
        if pitch % (4 * seg): continue
        total = rows * pitch
        for rpw, order, il in ((16, 0, 0), (4, 0, 0), (1, 0, 0), (1, 0, 2), (4, 0, 2), (16, 0, 2), (1, 1, 2), (16, 1, 0)):
            for nt in (0, 1):
                t = timeit(lambda: capi.check(lib.bsq_fill_pattern_device(buf.data_ptr(), rows, pitch, seg, rpw, order, il, nt, None)))
                print("pitch %8d (x%d) seg %5d rpw %2d order %d il %d nt %d -> %.4f ms %6.0f GB/s" % (pitch, mult, seg, rpw, order, il, nt, t, total / t / 1e6))
