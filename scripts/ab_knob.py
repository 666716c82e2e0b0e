#!/usr/bin/env python3
"""A/B a tuning knob on one shape, interleaved:  ab_knob.py SHAPE_INDEX PATH KNOB v1 v2 ...  (repeats 3 rounds)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioseq_amd import capi, synth
from sweep_shapes_list import SHAPES
lib = capi.load()
dev = torch.device("cuda:0")
si, path, knob = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
vals = [int(v) for v in sys.argv[4:]]
key, flags, B, lo, hi, P, dc = SHAPES[si]
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
capi.check(lib.bsq_tuning_set(b"onehot_path", path))
print(key, flags, B, P, dc, "rowbytes", C * sz, "path", path, "knob", knob)
for rnd in range(3):
    row = []
    for v in vals:
        capi.check(lib.bsq_tuning_set(knob.encode(), v))
        run(); torch.cuda.synchronize()
        ts = []
        if os.environ.get("AB_PER_STEP"):  # bench.py's protocol: one event pair per call, all queued before the sync
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(21)]
            for a, b in ev:
                a.record(); run(); b.record()
            torch.cuda.synchronize(); ts = [a.elapsed_time(b) for a, b in ev]
        else:
            for _ in range(7):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(3): run()
                b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 3)
        row.append("%d: %.3f ms (%.0f GB/s)" % (v, np.median(ts), algo / np.median(ts) / 1e6))
    print("  round %d  " % rnd + " | ".join(row), flush=True)
