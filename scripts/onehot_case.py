#!/usr/bin/env python3
"""One one-hot shape looped (for rocprofv3 --kernel-trace --stats: per-kernel durations of a shape the bench does not time by itself).
    onehot_case.py KEY eos bos pad B LEN P destchar [knob=value ...] [reps]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioseq_amd import capi, synth
lib = capi.load()
dev = torch.device("cuda:0")
key, eos, bos, pad, B, L, P, dc = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), sys.argv[8]
reps = 30
for a in sys.argv[9:]:
    if "=" in a:
        k, v = a.split("=")
        capi.check(lib.bsq_tuning_set(k.encode(), int(v)))
    else:
        reps = int(a)
chars, offs = synth.synth_packed(7, B, L, L, "ACGT" if key.startswith("DNA") else synth.AA)
desc = capi.make_desc(key, eos, bos, pad)
C = lib.bsq_alphabet_size(ctypes.byref(desc))
dt = ctypes.c_int(0); capi.check(lib.bsq_dtype_from_destchar(dc.encode(), ctypes.byref(dt)))
sz = lib.bsq_dtype_size(dt)
dch, dof = torch.from_numpy(chars).to(dev), torch.from_numpy(offs).to(dev)
out = torch.empty(P * B * C * sz, dtype=torch.uint8, device=dev)
for _ in range(reps):
    capi.check(lib.bsq_onehot_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), None, B, P, dt, out.data_ptr(), None))
torch.cuda.synchronize()
print("done", key, B, P, C, sz)
