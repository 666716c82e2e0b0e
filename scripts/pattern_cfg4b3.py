#!/usr/bin/env python3
"""Is the tile store pattern (1792-byte segments, 16 rows per wave) slow because too many stores are in flight?
Sweeps the workgroups per CU (unused LDS) and a per-wave store throttle (s_waitcnt) on the cfg4 int8 geometry."""
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
for seg, rpws in ((1792, (16, 4)), (4096, (1, 4)), (16384, (1,))):
    pitch = (7000000 // seg) * seg
    for rpw in rpws:
        for pad in (0, 20480, 32768, 40960, 53248, 65536):
            row = []
            for wait in (0, 1, 2, 3, 5):
                capi.check(lib.bsq_tuning_set(b"fill_pad", pad)); capi.check(lib.bsq_tuning_set(b"pattern_wait", wait))
                t = timeit(lambda: capi.check(lib.bsq_fill_pattern_device(buf.data_ptr(), 160, pitch, seg, rpw, 0, 0, 1, None)))
                row.append("w%d %.4f" % (wait, t))
            print("seg %5d rpw %2d pad %5d | %s" % (seg, rpw, pad, " | ".join(row)), flush=True)
