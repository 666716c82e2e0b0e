#!/usr/bin/env python3
"""Follow-up to pattern_cfg4b.py: does a LONGER contiguous segment per wave recover the linear-fill rate on the
cfg4 int8 geometry (160 rows x ~7 MB pitch)?  Segments of 1792 B .. 64 KiB, and the linear limit (one row)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioseq_amd import capi
lib = capi.load()
dev = torch.device("cuda:0")
def timeit(fn, n=10, reps=5):
    ts = []
    for _ in range(reps):
        fn(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / n)
    return float(np.median(ts))
buf = torch.empty(160 * 7100000 + 65536, dtype=torch.uint8, device=dev)
for seg in (1792, 4096, 7168, 8192, 14336, 16384, 28672, 65536):
    pitch = (7000000 // seg) * seg
    for rows, p in ((160, pitch), (1, 160 * pitch)):
        for rpw in (1, 2, 4):
            if rows == 1 and rpw > 1: continue
            for order in (0, 1):
                t = timeit(lambda: capi.check(lib.bsq_fill_pattern_device(buf.data_ptr(), rows, p, seg, rpw, order, 0, 1, None)))
                print("seg %5d rows %3d rpw %2d order %d -> %.4f ms %6.0f GB/s" % (seg, rows, rpw, order, t, rows * p / t / 1e6), flush=True)
capi.check(lib.bsq_fill_device(buf.data_ptr(), 1120000000, 7, None))
t = timeit(lambda: capi.check(lib.bsq_fill_device(buf.data_ptr(), 1120000000, 7, None)))
print("bsq_fill_device 1.12 GB -> %.4f ms %6.0f GB/s" % (t, 1120000000 / t / 1e6))
