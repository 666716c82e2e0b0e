#!/usr/bin/env python3
"""Store-pattern floor of a (P,B) int8 token matrix (cfg2 seq-first: 1024 rows x 65536 bytes): segment length
per tile row x rows per wave x workgroups per CU (bsq_fill_pattern_device; no work, stores only)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioseq_amd import capi
lib = capi.load()
dev = torch.device("cuda:0")
def timeit(fn, n=20, reps=7):
    ts = []
    for _ in range(reps):
        fn(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / n)
    return float(np.median(ts))
rows, pitch = 1024, 65536
buf = torch.empty(rows * pitch + 65536, dtype=torch.uint8, device=dev)
for seg in (256, 512, 1024, 2048, 4096):
    for rpw in (16, 4, 1):
        for order in (0, 1):
            row = []
            for pad in (0, 32768, 53248, 65536):
                capi.check(lib.bsq_tuning_set(b"fill_pad", pad))
                t = timeit(lambda: capi.check(lib.bsq_fill_pattern_device(buf.data_ptr(), rows, pitch, seg, rpw, order, 0, 1, None)))
                row.append("pad %5d %.1f us" % (pad, t * 1e3))
            print("seg %5d rpw %2d order %d | %s" % (seg, rpw, order, " | ".join(row)), flush=True)
