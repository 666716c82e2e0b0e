#!/usr/bin/env python3
"""Plain-fill write bandwidth vs resident workgroups per CU (unused dynamic LDS as the occupancy cap), plain and non-temporal stores."""
import os, sys; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from bioseq_amd import capi
lib = capi.load(); N = 5 * 1024**3
buf = torch.empty(N, dtype=torch.uint8, device="cuda")
for mode in (1, 3):  # 1: plain stores, 3: non-temporal stores; one 1-KiB store per wave, one aligned 4-KiB chunk per workgroup
  capi.check(lib.bsq_tuning_set(b"fill_mode", mode))
  for pad in (0, 8192, 16384, 20000, 32768, 40000, 53000, 65000):
      capi.check(lib.bsq_tuning_set(b"fill_pad", pad))
      ts = []
      for _ in range(4):
          capi.check(lib.bsq_fill_device(buf.data_ptr(), N, 0, None))
          a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
          a.record()
          for _ in range(5): capi.check(lib.bsq_fill_device(buf.data_ptr(), N, 0, None))
          b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 5)
      blocks = min(8, 163840 // max(pad, 1)) if pad else 8
      print("fill mode %d, LDS pad %6d (<= %d blocks/CU): %.4f ms %.0f GB/s" % (mode, pad, blocks, np.median(ts), N / np.median(ts) / 1e6))
