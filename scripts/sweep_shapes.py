#!/usr/bin/env python3
"""Compare the three one-hot paths (0 tile, 1 two-pass, 2 chunk-owner) over a set of shapes."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioseq_amd import capi, synth
lib = capi.load()
dev = torch.device("cuda:0")
from sweep_shapes_list import SHAPES
sel = sys.argv[1:] and [int(x) for x in sys.argv[1:]]
def setk(**kw):
    for k in ("nt_stores", "onehot_path"):
        capi.check(lib.bsq_tuning_set(k.encode(), int(kw.get(k, 0))))
for si, (key, flags, B, lo, hi, P, dc) in enumerate(SHAPES):
    if sel and si not in sel: continue
    letters = synth.AA if key[0] not in "D" else "ACGT"
    chars, offs = synth.synth_packed(1000 + si, B, lo, hi, letters)
    desc = capi.make_desc(key, *flags)
    C = lib.bsq_alphabet_size(ctypes.byref(desc))
    dt = ctypes.c_int(0); capi.check(lib.bsq_dtype_from_destchar(dc.encode(), ctypes.byref(dt)))
    sz = lib.bsq_dtype_size(dt)
    dch, dof = torch.from_numpy(chars).to(dev), torch.from_numpy(offs).to(dev)
    ob = P * B * C * sz
    out = torch.empty(ob, dtype=torch.uint8, device=dev); ref = torch.empty_like(out)
    algo = int(offs[-1]) + 8 * (B + 1) + ob
    def run(): capi.check(lib.bsq_onehot_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), None, B, P, dt, out.data_ptr(), None))
    res = []
    for path in (1, 2, 3, 0):  # 0 = the library's own choice, last: its time should match the best of the three
        for nt in (1,):
            setk(onehot_path=path, nt_stores=nt)
            out.fill_(5); run(); torch.cuda.synchronize()
            if path == 1: ref.copy_(out)
            else: assert torch.equal(out, ref), (key, path)
            ts = []
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(3): run()
                b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 3)
            res.append("%s %.3f ms %4.0f GB/s" % ("auto" if path == 0 else "p%d" % path, np.median(ts), algo / np.median(ts) / 1e6))
    print("%-8s %s B=%7d P=%4d %s C=%2d rowbytes=%3d out=%5.2f GB pitch%%32K=%5d | %s" % (key, flags, B, P, dc, C, C * sz, ob / 1e9, (B * C * sz) % 32768, " | ".join(res)), flush=True)
    del out, ref, dch, dof
setk()
