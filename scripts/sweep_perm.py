#!/usr/bin/env python3
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioseq_amd import capi
lib = capi.load()
N = 5 * 1024 * 1024 * 1024
buf = torch.empty(N + (1 << 20), dtype=torch.uint8, device="cuda:0")
def timeit(fn, n=5, reps=4):
    ts = []
    for _ in range(reps):
        fn(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / n)
    return float(np.median(ts))
names = {0: "identity", 2: "rotate class per group of 8", 3: "swap neighbours", 4: "bit-reverse in 64-windows", 5: "constant class rotation", 6: "8x8 transpose in 64-windows"}
for seg in (1024, 256, 512):
    for order in (0, 5, 3, 2, 4, 6):
        t = timeit(lambda: capi.check(lib.bsq_fill_pattern_device(buf.data_ptr(), 1, N, seg, 1, order, 2, 0, None)))
        print("block = %5d B contiguous, map %-30s -> %.4f ms %6.0f GB/s" % (4 * seg, names[order], t, N / t / 1e6))
