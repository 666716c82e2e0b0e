#!/usr/bin/env python3
"""Occupancy sweep over shapes: two-pass (expansion kernel, pad via expand_pad) and chunk-owner
(pad via chunks_pad) with the unused-LDS pads that give 8/7/6/5/4/3 resident workgroups per CU."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioseq_amd import capi, synth
from sweep_shapes_list import SHAPES
lib = capi.load()
dev = torch.device("cuda:0")
PADS = [(8, 1024), (6, 10240), (5, 15360), (4, 22528), (3, 36864)]
sel = sys.argv[1:] and [int(x) for x in sys.argv[1:]]
def setk(**kw):
    for k in ("onehot_path", "expand_pad", "chunks_pad"):
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
    out = torch.empty(ob, dtype=torch.uint8, device=dev)
    algo = int(offs[-1]) + 8 * (B + 1) + ob
    def run(): capi.check(lib.bsq_onehot_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), None, B, P, dt, out.data_ptr(), None))
    res = []
    for path in (2, 3):
        row = []
        for wg, pad in PADS:
            if path == 2: setk(onehot_path=2, expand_pad=pad)
            else: setk(onehot_path=3, chunks_pad=pad)
            run(); torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(3): run()
                b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 3)
            row.append("%d:%.3f" % (wg, np.median(ts)))
        res.append("p%d " % path + " ".join(row))
    print("%-8s %s B=%7d P=%4d %s rowbytes=%3d out=%5.2f GB | %s" % (key, flags, B, P, dc, C * sz, ob / 1e9, " | ".join(res)), flush=True)
    del out, dch, dof
setk()
