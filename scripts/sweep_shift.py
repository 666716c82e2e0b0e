#!/usr/bin/env python3
"""Does the fast 'one 1-KiB store per wave, 4 KiB per block' fill depend on WHICH 4-KiB chunk a block
(= an XCD, blocks are dealt round-robin over the 8 XCDs) writes?  Shift the base by s chunks."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioseq_amd import capi
lib = capi.load()
dev = torch.device("cuda:0")
N = 5 * 1024 * 1024 * 1024
buf = torch.empty(N + (1 << 20), dtype=torch.uint8, device=dev)

def timeit(fn, n=5, reps=4):
    ts = []
    for _ in range(reps):
        fn(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / n)
    return float(np.median(ts))

capi.check(lib.bsq_tuning_set(b"fill_mode", 1))
for shift in (0, 1024, 2048, 4096, 8192, 12288, 16384, 20480, 24576, 28672, 32768, 65536, 4096 * 3 + 1024):
    t = timeit(lambda: capi.check(lib.bsq_fill_device(buf.data_ptr() + shift, N, 0, None)))
    print("fill_mode 1, base shift %6d B -> %.4f ms %6.0f GB/s" % (shift, t, N / t / 1e6))
# one row of N bytes; seg bytes per wave; rowwise pattern (4 waves of a block adjacent)
for seg in (1024, 2048, 4096, 8192):
    t = timeit(lambda: capi.check(lib.bsq_fill_pattern_device(buf.data_ptr(), 1, N, seg, 1, 0, 2, 0, None)))
    print("rowwise single row seg %5d (block = %6d B) -> %.4f ms %6.0f GB/s" % (seg, 4 * seg, t, N / t / 1e6))
capi.check(lib.bsq_tuning_set(b"fill_mode", 0))
