#!/usr/bin/env python3
"""Round 3: the STORE PATTERN of the fused chunk-column one-hot kernel designed for cfg4 int8 (a
workgroup owns R position rows x the aligned 4-KiB chunks of ~4096 sequences, tokens in LDS, every wave writes one chunk
(or 2 / 7 contiguous ones) in each of its rows) -- stores only, none of the kernel's work.  If the pattern alone cannot
beat the tiled kernel's 0.22 ms by a margin, the kernel cannot either.  Geometry: 160 rows of ~7 MB (1.12 GB)."""
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
print("rows 160; interleave 2 = the 4 waves of a block write 4 adjacent segments of the same row, rpw rows per block")
for seg in (4096, 8192, 28672):
    pitch = (7000000 // (4 * seg)) * (4 * seg)
    nbytes = 160 * pitch
    for rpw in (1, 4, 8, 16, 32):
        row = []
        for pad in (0, 40960, 65536):   # unused LDS: 8 / 3-4 / 2 blocks per CU
            for order in (0, 1):
                capi.check(lib.bsq_tuning_set(b"fill_pad", pad)); capi.check(lib.bsq_tuning_set(b"pattern_wait", 0))
                t = timeit(lambda: capi.check(lib.bsq_fill_pattern_device(buf.data_ptr(), 160, pitch, seg, rpw, order, 2, 1, None)))
                row.append("pad %5d ord %d: %.4f ms (%.2f TB/s)" % (pad, order, t, nbytes / t / 1e9))
        print("seg %5d (wave writes %d chunk(s) per row) x %2d rows per wave | %s" % (seg, seg // 4096, rpw, " | ".join(row)), flush=True)
capi.check(lib.bsq_tuning_set(b"fill_pad", 0))
