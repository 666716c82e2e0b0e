#!/usr/bin/env python3
"""k_expand_rows1 (round 5: the LDS-free expansion of one-byte rows of 3 ... 15 bytes) against k_expand_chunks and the tiled kernel:
int8 one-hot shapes with tiny rows (cfg4b, its 1/8 shard, DNA5 / DNA4 without specials, 14-byte protein alphabets), every
(onehot_path, expand_rows1, expand_pad) arm checked against the tiled kernel's output (itself oracle-checked by the GPU suite) first.
    rows1_lab.py [pads]        pads: comma list of expand_pad values for the two-pass arms (default 0)"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioseq_amd import capi, synth
lib = capi.load()
dev = torch.device("cuda:0")
pads = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0]
SHAPES = [("DNA4", (1, 1, 1), 1000000, 150, 150, 160), ("DNA4", (1, 1, 1), 125000, 150, 150, 160), ("DNA4", (1, 1, 1), 250000, 150, 150, 160),
          ("DNA5", (0, 0, 0), 131072, 50, 1024, 1024), ("DNA4", (0, 0, 0), 262144, 30, 512, 512), ("DNA4", (1, 1, 1), 65536, 50, 2048, 2048),
          ("SEB14", (0, 0, 0), 131072, 30, 512, 512), ("SEB8", (1, 1, 1), 262144, 30, 512, 512), ("DNA4", (1, 1, 1), 250001, 100, 256, 256),
          ("DAYHOFF", (0, 0, 0), 100000, 30, 300, 300), ("DNA4", (1, 1, 1), 16384, 100, 256, 256)]
def setk(**kw):
    for k in ("onehot_path", "expand_rows1", "expand_pad"):
        capi.check(lib.bsq_tuning_set(k.encode(), int(kw.get(k, 0))))
for si, (key, flags, B, lo, hi, P) in enumerate(SHAPES):
    letters = synth.AA if key[0] != "D" or key == "DAYHOFF" else "ACGT"
    chars, offs = synth.synth_packed(2000 + si, B, lo, hi, letters)
    desc = capi.make_desc(key, *flags)
    C = lib.bsq_alphabet_size(ctypes.byref(desc))
    dt = ctypes.c_int(0); capi.check(lib.bsq_dtype_from_destchar(b"B", ctypes.byref(dt)))
    dch, dof = torch.from_numpy(chars).to(dev), torch.from_numpy(offs).to(dev)
    ob = P * B * C
    buf = torch.empty(ob + 4096, dtype=torch.uint8, device=dev); ref = torch.empty(ob, dtype=torch.uint8, device=dev)
    algo = int(offs[-1]) + 8 * (B + 1) + ob
    res = []
    for shift in (0, 16):   # the result at a 4-KiB boundary and 16 bytes off it (chunks then straddle differently; head != 0)
        out = buf[shift:shift + ob]
        def run(): capi.check(lib.bsq_onehot_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), None, B, P, dt, out.data_ptr(), None))
        arms = [("tile", dict(onehot_path=1))] + [("2p-lds pad%d" % pd, dict(onehot_path=2, expand_rows1=1, expand_pad=pd)) for pd in pads[:1]] + \
               [("2p-rows1 pad%d" % pd, dict(onehot_path=2, expand_rows1=0, expand_pad=pd)) for pd in pads] + [("auto", dict())]
        for name, kw in arms:
            setk(**kw)
            out.fill_(5); run(); torch.cuda.synchronize()
            if name == "tile" and shift == 0: ref.copy_(out)
            else: assert torch.equal(out, ref), (key, name, shift)
            ts = []
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(5): run()
                b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 5)
            res.append("%s%s %.1f us %4.0f GB/s" % (name, "+16" if shift else "", np.median(ts) * 1e3, algo / np.median(ts) / 1e6))
    print("%-8s %s B=%7d P=%4d C=%2d out=%5.2f GB | %s" % (key, flags, B, P, C, ob / 1e9, " | ".join(res)), flush=True)
    del buf, ref, dch, dof
setk()
