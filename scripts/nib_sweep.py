#!/usr/bin/env python3
"""Nibble id scratch over the shapes it applies to (alphabets of <= 15 classes, elements of 2 bytes and more, rows of >= 16 bytes, results
>= 192 MB): byte ids (knob raw_nibbles = 1) at the default occupancy pad against nibbles at several pads; every arm bit-identical to the
byte arm's output (itself oracle-checked by the GPU suite).      [PADS=0,4096,...] nib_sweep.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioseq_amd import capi, synth
lib = capi.load()
dev = torch.device("cuda:0")
pads = [int(x) for x in os.environ.get("PADS", "0,4096,8192,12288").split(",")]
SHAPES = [("DNA4", (1, 1, 1), 1000000, 150, 150, 160, b"f"), ("DNA4", (0, 0, 0), 262144, 30, 512, 512, b"f"), ("DNA5", (1, 1, 1), 500000, 100, 254, 256, b"f"),
          ("SEB8", (1, 1, 1), 131072, 30, 510, 512, b"f"), ("DAYHOFF", (0, 0, 0), 262144, 30, 512, 512, b"f"), ("LIA10", (1, 1, 1), 131072, 30, 510, 512, b"f"),
          ("DNA4", (1, 1, 1), 500000, 150, 150, 160, b"d"), ("SEB8", (0, 0, 0), 131072, 30, 512, 512, b"d"), ("LIA10", (1, 1, 1), 65536, 30, 510, 512, b"d"),
          ("DNA5", (1, 1, 1), 500000, 100, 254, 256, b"h"), ("DNA4", (1, 1, 1), 125000, 150, 150, 160, b"f")]
if os.environ.get("SET") == "2":   # rows of 24 ... 31 bytes, where nibbles won in the first sweep: other batch shapes, other element types
    SHAPES = [("DNA4", (1, 1, 1), 262144, 30, 510, 512, b"f"), ("DNA4", (1, 1, 1), 65536, 50, 2046, 2048, b"f"), ("DNA4", (1, 1, 1), 250001, 100, 254, 256, b"f"),
              ("DNA5", (1, 1, 0), 500000, 100, 254, 256, b"f"), ("DAYHOFF", (0, 0, 0), 1000000, 100, 150, 160, b"f"), ("DAYHOFF", (1, 0, 0), 262144, 30, 511, 512, b"f"),
              ("SEB14", (0, 0, 0), 131072, 30, 512, 512, b"h"), ("LIA10", (1, 1, 1), 262144, 30, 510, 512, b"h"), ("SEB10", (1, 1, 1), 262144, 30, 510, 512, b"h"),
              ("DNA4", (1, 1, 1), 2000000, 150, 150, 160, b"f"), ("DNA4", (1, 1, 1), 500000, 150, 150, 160, b"f")]
if os.environ.get("SET") == "3":   # id matrices beyond the Infinity Cache (320 MB as bytes), other row widths
    SHAPES = [("DNA5", (1, 1, 1), 2000000, 150, 150, 160, b"f"), ("DNA4", (0, 0, 0), 2000000, 150, 160, 160, b"f"), ("SEB8", (1, 1, 1), 2000000, 100, 158, 160, b"f"),
              ("DNA4", (1, 1, 1), 1500000, 150, 150, 160, b"f"), ("DNA4", (1, 1, 1), 1250000, 150, 150, 160, b"f")]
if os.environ.get("SET") == "4":   # alphabets of more than 15 classes (bytes only) with id matrices around and beyond the Infinity Cache
    SHAPES = [("AMINO20", (0, 0, 0), 65536, 50, 1024, 1024, b"B"), ("AMINO20", (0, 0, 0), 131072, 50, 1024, 1024, b"B"), ("AMINO20", (0, 0, 0), 196608, 50, 1024, 1024, b"B"),
              ("AMINO20", (0, 0, 0), 262144, 50, 1024, 1024, b"B"), ("AMINO20", (0, 0, 0), 131072, 50, 1024, 1024, b"f"), ("AMINO20", (0, 0, 0), 262144, 50, 1024, 1024, b"f")]
def setk(nib, pad):
    capi.check(lib.bsq_tuning_set(b"raw_nibbles", nib)); capi.check(lib.bsq_tuning_set(b"expand_pad", pad))
for si, (key, flags, B, lo, hi, P, dc) in enumerate(SHAPES):
    letters = synth.AA if key[0] != "D" or key == "DAYHOFF" else "ACGT"
    chars, offs = synth.synth_packed(3000 + si, B, lo, hi, letters)
    desc = capi.make_desc(key, *flags)
    C = lib.bsq_alphabet_size(ctypes.byref(desc))
    dt = ctypes.c_int(0); capi.check(lib.bsq_dtype_from_destchar(dc, ctypes.byref(dt)))
    sz = lib.bsq_dtype_size(dt)
    dch, dof = torch.from_numpy(chars).to(dev), torch.from_numpy(offs).to(dev)
    ob = P * B * C * sz
    out = torch.empty(ob, dtype=torch.uint8, device=dev); ref = torch.empty(ob, dtype=torch.uint8, device=dev)
    algo = int(offs[-1]) + 8 * (B + 1) + ob
    capi.check(lib.bsq_tuning_set(b"onehot_path", 2))
    def run(): capi.check(lib.bsq_onehot_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), None, B, P, dt, out.data_ptr(), None))
    arms = [("bytes", 1, 0)] + [("nibbles pad%d" % pd, 0, pd) for pd in pads]
    res = {n: [] for n, _, _ in arms}
    for n, nib, pd in arms:
        setk(nib, pd); out.fill_(3); run(); torch.cuda.synchronize()
        if n == "bytes": ref.copy_(out)
        else: assert torch.equal(out, ref), (key, n)
    for rnd in range(3):
        for n, nib, pd in arms:
            setk(nib, pd)
            for _ in range(5): run()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): run()
            b.record(); torch.cuda.synchronize(); res[n].append(a.elapsed_time(b) / 10)
    print("%-8s %s B=%7d P=%4d C=%2d sz=%d row=%3d B out=%5.2f GB | %s" % (key, flags, B, P, C, sz, C * sz, ob / 1e9,
          " | ".join("%s %.1f us %.3f" % (n, np.median(res[n]) * 1e3, algo / np.median(res[n]) / 8e9) for n, _, _ in arms)), flush=True)
    del out, ref, dch, dof
setk(0, 0); capi.check(lib.bsq_tuning_set(b"onehot_path", 0))
