#!/usr/bin/env python3
"""bsq_augment_tokenize_device over batch sizes (cfg5's shape: SEB8, len~U(30,512), padlen 512, int8 (B,P)): the one-launch form
(knob augment_fused 0) against the two launches (1), resident loop.    aug_sizes_lab.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioseq_amd import capi, synth
lib = capi.load()
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
c = synth.CONFIGS["cfg5"]
desc = capi.make_desc(c["key"], c["eos"], c["bos"], c["padchar"])
P = c["padlen"]
for n in [int(x) for x in os.environ.get("SIZES", "1024,4096,8192,16384,32768,65536,131072,262144").split(",")]:
    chars, offs = synth.synth_packed(c["seed"], n, c["lo"], c["hi"], c["letters"])
    dch, dof = torch.from_numpy(chars).to(dev), torch.from_numpy(offs).to(dev)
    pristine = dch.clone()
    out = torch.empty((n, P), dtype=torch.int8, device=dev)
    res = []
    for knob in [int(x) for x in os.environ.get("KNOBS", "0,1,2").split(",")]:
        capi.check(lib.bsq_tuning_set(b"augment_fused", knob))
        seed = [0]
        def step():
            seed[0] += 1
            capi.check(lib.bsq_augment_tokenize_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), n, P, 1, capi.I8, out.data_ptr(), int(os.environ.get("CHAIN", "1")), float(os.environ.get("FRAC", "0.5")), seed[0], stream.cuda_stream))
        ts = []
        for rep in range(5):
            dch.copy_(pristine)
            for _ in range(20): step()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(50): step()
            b.record(stream); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 50 * 1e3)
        res.append("knob %d: %.2f us" % (knob, float(np.median(ts))))
    capi.check(lib.bsq_tuning_set(b"augment_fused", 0))
    capi.check(lib.bsq_fused_status(None))
    print("B = %6d  %s" % (n, " | ".join(res)), flush=True)
