#!/usr/bin/env python3
"""One-hot with a mask (the masked-LM leg of the reference's training loop, training/cnnpretrain.py:124) against the
unmasked call, on the cfg3 / cfg4 batches: f32 and int8, (P,B,C) and channels-first.  C ABI, event-timed."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioseq_amd import capi, synth
lib = capi.load()
dev = torch.device("cuda:0")
def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): fn()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 5)
    return float(np.median(ts))
buf = torch.empty(6 * 2**30, dtype=torch.uint8, device=dev)
for name in ("cfg3", "cfg4"):
    c = synth.CONFIGS[name]
    chars, offs = synth.synth_packed(c["seed"], c["n"], c["lo"], c["hi"], c["letters"])
    mask = (np.random.default_rng(1).random(chars.size) < 0.85).astype(np.uint8)
    dch, dof, dm = torch.from_numpy(chars).to(dev), torch.from_numpy(offs).to(dev), torch.from_numpy(mask).to(dev)
    desc = capi.make_desc(c["key"], c["eos"], c["bos"], c["padchar"])
    B, P = c["n"], c["padlen"]
    C = lib.bsq_alphabet_size(ctypes.byref(desc))
    for dt, dname, sz in ((capi.F32, "f32", 4), (capi.I8, "int8", 1)):
        algo = int(offs[-1]) + 8 * (B + 1) + P * B * C * sz
        for entry, lname in ((lib.bsq_onehot_device, "(P,B,C)"), (lib.bsq_onehot_bcl_device, "(B,C,P)")):
            row = []
            for m in (None, dm.data_ptr()):
                t = timeit(lambda: capi.check(entry(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), m, B, P, dt, buf.data_ptr(), None)))
                a = algo + (int(offs[-1]) if m else 0)
                row.append("%s %.3f ms %.2f TB/s" % ("masked" if m else "plain ", t, a / t / 1e9))
            print("%s %-4s %s | %s" % (name, dname, lname, " | ".join(row)), flush=True)
    pitch = (B + 255) // 256 * 256
    row = []
    for m in (None, dm.data_ptr()):
        t = timeit(lambda: capi.check(lib.bsq_raw_tokens_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), m, B, P, buf.data_ptr(), pitch, None)))
        row.append("%s %.3f ms" % ("masked" if m else "plain ", t))
    print("%s token pass alone (k_tokens_raw) | %s" % (name, " | ".join(row)), flush=True)
