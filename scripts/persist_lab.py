#!/usr/bin/env python3
"""The read + write stream of the (B,P) token kernels' shape (bsq_copy_mix_device) in its forms, on a resident buffer pair and COLD
(cycling over > 512 MiB of distinct pairs): mode 0 / 1 one chunk per wave (one / two dependent load steps), mode 4 persistent waves with
the next chunk's loads issued before the current stores, modes 5 / 6 / 7 an LDS-DMA loader wave 1 / 2 / 3 steps ahead of three consumer
waves.  Shapes: cfg2 (64 MiB out, 35 MB in) and cfg5 (128 MiB out, 71 MB in).    persist_lab.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioseq_amd import capi
lib = capi.load()
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
sh = stream.cuda_stream

def timed(fn, n, warm):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(n): fn()
    b.record(stream); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for name, out_bytes, in_bytes in (("cfg2", 64 << 20, 35121248 // 16 * 16), ("cfg5", 128 << 20, 70957455 // 16 * 16)):
    nb = max(8, -(-(513 << 20) // in_bytes))
    srcs = [torch.randint(0, 255, (in_bytes,), dtype=torch.uint8, device=dev) for _ in range(nb)]
    dsts = [torch.empty(out_bytes, dtype=torch.uint8, device=dev) for _ in range(nb)]
    it = [0]
    def run(mode, cold):
        k = it[0] % nb if cold else 0
        it[0] += 1
        capi.check(lib.bsq_copy_mix_device(dsts[k].data_ptr(), out_bytes, srcs[k].data_ptr(), in_bytes, mode, 1, sh))
    # correctness of the new forms: same bytes as mode 0
    ref = torch.empty_like(dsts[0])
    capi.check(lib.bsq_copy_mix_device(ref.data_ptr(), out_bytes, srcs[0].data_ptr(), in_bytes, 0, 1, sh))
    per_chunk = -(-(in_bytes // 16) // (out_bytes // 4096))
    i16 = torch.arange(out_bytes // 16, device=dev)
    live = ((i16 % 256 < per_chunk) & ((i16 // 256) * per_chunk + i16 % 256 < in_bytes // 16)).repeat_interleave(16)   # pieces that hold loaded data ...
    live &= (torch.arange(out_bytes, device=dev) % 16 >= 4)   # ... but for their first word (or-ed with a neighbour piece that may be a filler)
    for mode in (4, 5, 6, 7):
        for wg in (2, 4):
            capi.check(lib.bsq_tuning_set(b"fill_mode", wg))
            dsts[0].fill_(0)
            run(mode, False); torch.cuda.synchronize()
            assert torch.equal(dsts[0][live], ref[live]), (name, mode, wg)
    arms = [(0, 0, 0), (1, 0, 0), (0, 0, 53000), (1, 0, 53000)] + [(4, w, 0) for w in (1, 2, 3, 4, 6, 8)] + [(m, w, 0) for m in (5, 6, 7) for w in (1, 2, 3, 4)]
    for rnd in range(2):
        for mode, wg, pad in arms:
            capi.check(lib.bsq_tuning_set(b"fill_mode", wg))
            capi.check(lib.bsq_tuning_set(b"fill_pad", pad))
            res = timed(lambda: run(mode, False), 300, 100)
            cold = timed(lambda: run(mode, True), 40 * nb, 2 * nb)
            tot = out_bytes + in_bytes
            print("%s round %d mode %d wgs/CU %d pad %5d | resident %6.2f us %5.0f GB/s | cold %6.2f us %5.0f GB/s (frac of 8 TB/s %.3f)" % (
                name, rnd, mode, wg, pad, res, tot / res / 1e3, cold, tot / cold / 1e3, tot / cold / 1e3 / 8000), flush=True)
    capi.check(lib.bsq_tuning_set(b"fill_mode", 0)); capi.check(lib.bsq_tuning_set(b"fill_pad", 0))
    del srcs, dsts, ref
    torch.cuda.empty_cache()
