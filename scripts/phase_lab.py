#!/usr/bin/env python3
"""Does HBM prefer its reads and writes in separate phases?  The (B,P) token stream of cfg2 / cfg5 in the COLD regime (cycling over
buffers that add up to > 512 MiB of input): the copy-mix yardstick (reads and writes interleaved, as the token kernels issue them)
against its read half alone, its write half alone (a fill), and the two run back to back as two launches."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioseq_amd import capi
lib = capi.load()
dev = torch.device("cuda:0")
capi.check(lib.bsq_tuning_set(b"fill_mode", 1))
for name, src_b, dst_b in (("cfg2", 35121248 // 16 * 16, 67108864), ("cfg5", 70957455 // 16 * 16, 134217728)):
    nb = max(8, -(-(513 << 20) // src_b))
    srcs = [torch.randint(0, 255, (src_b,), dtype=torch.uint8, device=dev) for _ in range(nb)]
    dsts = [torch.empty(dst_b, dtype=torch.uint8, device=dev) for _ in range(nb)]
    it = [0]
    def mix(mode):
        def f():
            k = it[0] % nb; it[0] += 1
            capi.check(lib.bsq_copy_mix_device(dsts[k].data_ptr(), dst_b, srcs[k].data_ptr(), src_b, mode, 1, None))
        return f
    def fill():
        k = it[0] % nb; it[0] += 1
        capi.check(lib.bsq_fill_device(dsts[k].data_ptr(), dst_b, 0, None))
    def read_then_fill():
        k = it[0] % nb; it[0] += 1
        capi.check(lib.bsq_copy_mix_device(dsts[k].data_ptr(), dst_b, srcs[k].data_ptr(), src_b, 3, 1, None))
        capi.check(lib.bsq_fill_device(dsts[k].data_ptr(), dst_b, 0, None))
    def loop_us(fn, n=2000):
        for _ in range(200): fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e3
    for rnd in range(2):
        r = {"mix, dependent loads (the yardstick)": loop_us(mix(1)), "mix, one load step": loop_us(mix(0)), "loads only": loop_us(mix(3)),
             "stores only (fill)": loop_us(fill), "loads, then fill (two launches)": loop_us(read_then_fill)}
        print(name, "%d buffers" % nb, " | ".join("%s %.2f us" % kv for kv in r.items()), flush=True)
        print("   read %.2f TB/s alone, write %.2f TB/s alone, mixed %.2f TB/s; loads + stores alone = %.2f us" % (
            src_b / r["loads only"] / 1e6, dst_b / r["stores only (fill)"] / 1e6, (src_b + dst_b) / r["mix, one load step"] / 1e6,
            r["loads only"] + r["stores only (fill)"]), flush=True)
    del srcs, dsts
    torch.cuda.empty_cache()
