#!/usr/bin/env python3
"""Performance cliffs: every entry point on a friendly shape and on its awkward neighbour (batch size / padlen that
is not a multiple of 16, odd sizes, misaligned output views).  Prints us per call and GB/s (algorithmic bytes)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioseq_amd import capi, synth
lib = capi.load()
dev = torch.device("cuda:0")
def timeit(fn):
    for _ in range(5): fn()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): fn()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 10)
    return float(np.median(ts))
def batch(key, B, lo, hi):
    chars, offs = synth.synth_packed(11, B, lo, hi, synth.AA if key[0] in "AS" else "ACGT")
    return torch.from_numpy(chars).to(dev), torch.from_numpy(offs).to(dev), int(offs[-1])
CASES = []
for key, flags, Bs, Ps in (("AMINO20", (0, 0, 0), (65536, 65000, 65537), (1024, 1000, 1001)),
                           ("DNA", (1, 1, 1), (262144, 250000, 250001), (256, 250, 251))):
    for B in Bs:
        for P in Ps:
            CASES.append((key, flags, B, P))
buf = torch.empty(6 * 2**30, dtype=torch.uint8, device=dev)
for key, flags, B, P in CASES:
    desc = capi.make_desc(key, *flags)
    C = lib.bsq_alphabet_size(ctypes.byref(desc))
    dch, dof, nch = batch(key, B, 30, P - 2)
    row = []
    for what, dc, bf, shift in (("tok", "b", 1, 0), ("tok", "b", 0, 0), ("tok", "b", 1, 1), ("tok", "i", 1, 0), ("tok", "i", 0, 0), ("tok", "d", 0, 0),
                                ("hot", "b", 0, 0), ("hot", "b", 0, 1), ("hot", "f", 0, 0), ("hot", "f", 0, 4), ("bcl", "f", 0, 0)):
        dt = ctypes.c_int(0); capi.check(lib.bsq_dtype_from_destchar(dc.encode(), ctypes.byref(dt)))
        sz = lib.bsq_dtype_size(dt)
        ob = B * P * sz * (1 if what == "tok" else C)
        if ob + 64 > buf.numel(): row.append("%s-%s skip" % (what, dc)); continue
        out = buf[shift:shift + ob]
        if what == "tok":
            fn = lambda: capi.check(lib.bsq_tokenize_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), B, P, bf, dt, out.data_ptr(), None))
            name = "tok-%s-%s" % (dc, "BP" if bf else "PB")
        elif what == "hot":
            fn = lambda: capi.check(lib.bsq_onehot_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), None, B, P, dt, out.data_ptr(), None))
            name = "hot-%s" % dc
        else:
            fn = lambda: capi.check(lib.bsq_onehot_bcl_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), None, B, P, dt, out.data_ptr(), None))
            name = "bcl-%s" % dc
        if shift: name += "+%d" % shift
        ms = timeit(fn)
        row.append("%s %.0f us %.1f TB/s" % (name, ms * 1e3, (nch + 8 * (B + 1) + ob) / ms / 1e9))
    print("%-7s B=%6d P=%4d | %s" % (key, B, P, " | ".join(row)), flush=True)
