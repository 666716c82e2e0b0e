#!/usr/bin/env python3
"""The tiled one-hot kernel's STORE PATTERN on the cfg4 int8 geometry (160 rows x ~7 MB pitch, 1792-byte segments),
without any of its work: how much do fewer rows per wave buy?  (diagnostic; bsq_fill_pattern_device)"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioseq_amd import capi
lib = capi.load()
dev = torch.device("cuda:0")
rows = 160
def timeit(fn, n=10, reps=5):
    ts = []
    for _ in range(reps):
        fn(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / n)
    return float(np.median(ts))
for seg in (1792, 3584, 896):
    pitch = (7000000 // seg) * seg
    buf = torch.empty(rows * pitch + 4096, dtype=torch.uint8, device=dev)
    total = rows * pitch
    for rpw in (16, 8, 4, 2, 1):
        for order in (0, 1):
            for il in (0, 2):
                if il == 2 and (pitch // seg) % 4: continue
                t = timeit(lambda: capi.check(lib.bsq_fill_pattern_device(buf.data_ptr(), rows, pitch, seg, rpw, order, il, 1, None)))
                print("seg %5d rpw %2d order %d il %d -> %.4f ms %6.0f GB/s" % (seg, rpw, order, il, t, total / t / 1e6), flush=True)
    del buf
