#!/usr/bin/env python3
"""k_expand_cmp (LDS-free compare form of the (P,B,C) expansion, knob expand_mode=5) vs k_expand_chunks: equality on odd
shapes, then timings over occupancy caps on cfg3 / cfg4 f32 / AMINO20 int8.  Run on the GPU box."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioseq_amd import capi, synth
lib = capi.load(); dev = torch.device("cuda:0")
def setk(**kw):
    for k, v in kw.items(): capi.check(lib.bsq_tuning_set(k.encode(), v))
def onehot(desc, dch, dof, B, P, dt, out):
    capi.check(lib.bsq_onehot_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), None, B, P, dt, out.data_ptr(), None))
ok = True
for key, flags, B, lo, hi, P, dc in [("AMINO20", (1, 1, 1), 3000, 0, 298, 300, "f"), ("DNA", (1, 1, 1), 70001, 1, 150, 152, "f"),
                                      ("SEB8", (0, 0, 0), 513, 0, 64, 64, "d"), ("AMINO20", (0, 0, 0), 4099, 0, 77, 77, "b"),
                                      ("DNA5", (1, 0, 1), 9, 0, 5000, 5002, "f"), ("AMINO20", (0, 1, 0), 2, 0, 3, 4, "h")]:
    chars, offs = synth.synth_packed(B + P, B, lo, hi, synth.DIRTY)
    desc = capi.make_desc(key, *flags); C = lib.bsq_alphabet_size(ctypes.byref(desc))
    dt = ctypes.c_int(0); capi.check(lib.bsq_dtype_from_destchar(dc.encode(), ctypes.byref(dt))); sz = lib.bsq_dtype_size(dt)
    dch = torch.from_numpy(np.concatenate([chars, np.zeros(1, np.uint8)])).to(dev)[:len(chars)]; dof = torch.from_numpy(offs).to(dev)
    a = torch.full((P * B * C * sz,), 7, dtype=torch.uint8, device=dev); b = torch.full_like(a, 9)
    setk(onehot_path=2, expand_mode=0); onehot(desc, dch, dof, B, P, dt, a)
    setk(expand_mode=5); onehot(desc, dch, dof, B, P, dt, b); torch.cuda.synchronize()
    same = bool(torch.equal(a, b)); ok &= same
    print("equal" if same else "MISMATCH", key, flags, B, P, dc, "C", C, flush=True)
print("correctness:", "OK" if ok else "FAILED")
def timeit(fn, n=30, warm=15):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
for name, dc in (("cfg3", "f"), ("cfg4", "f"), ("cfg3", "b"), ("cfg3", "d")):
    cfg = synth.CONFIGS[name]; B, P = cfg["n"], cfg["padlen"]
    chars, offs = synth.synth_packed(cfg["seed"], B, cfg["lo"], cfg["hi"], cfg["letters"])
    desc = capi.make_desc(cfg["key"], cfg["eos"], cfg["bos"], cfg["padchar"]); C = lib.bsq_alphabet_size(ctypes.byref(desc))
    dt = ctypes.c_int(0); capi.check(lib.bsq_dtype_from_destchar(dc.encode(), ctypes.byref(dt))); sz = lib.bsq_dtype_size(dt)
    dch, dof = torch.from_numpy(chars).to(dev), torch.from_numpy(offs).to(dev)
    out = torch.empty(P * B * C * sz, dtype=torch.uint8, device=dev)
    algo = len(chars) + 8 * (B + 1) + out.numel()
    fn = lambda: onehot(desc, dch, dof, B, P, dt, out)
    for rnd in range(2):
        for mode, pads in ((0, (0,)), (5, (0, 36864, 24576, 16384, -1))):
            for pad in pads:
                setk(onehot_path=2, expand_mode=mode, expand_pad=pad)
                t = timeit(fn)
                print("%s %s rows %3d B  expand_mode=%d expand_pad=%6d: %.4f ms  %.0f GB/s (%.3f)" % (name, dc, C * sz, mode, pad, t, algo / t / 1e6, algo / t / 1e6 / 8000), flush=True)
    del out
setk(onehot_path=0, expand_mode=0, expand_pad=0)
