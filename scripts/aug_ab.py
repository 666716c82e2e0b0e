#!/usr/bin/env python3
"""cfg5 + BLOSUM62 augmentation: the forms of bsq_augment_tokenize_device interleaved on one box (knob augment_fused),
each checked against the two-call form first.   aug_ab.py [knob values ...]   (default 0 1)"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioseq_amd import capi, synth
lib = capi.load()
dev = torch.device("cuda:0")
vals = [int(v) for v in sys.argv[1:]] or [0, 1]
cfg = synth.CONFIGS[os.environ.get("AUG_CFG", "cfg5")]
n, P = cfg["n"], cfg["padlen"]
chars, offs = synth.synth_packed(cfg["seed"], n, cfg["lo"], cfg["hi"], cfg["letters"])
desc = capi.make_desc(cfg["key"], cfg["eos"], cfg["bos"], cfg["padchar"])
pristine = torch.from_numpy(chars).to(dev)
dof = torch.from_numpy(offs).to(dev)
out = torch.empty((n, P), dtype=torch.int8, device=dev)
algo = int(offs[-1]) + 8 * (n + 1) + n * P
chain, frac = int(os.environ.get("AUG_CHAIN", "1")), float(os.environ.get("AUG_FRAC", "0.5"))

def step(buf, seed):
    capi.check(lib.bsq_augment_tokenize_device(ctypes.byref(desc), buf.data_ptr(), dof.data_ptr(), n, P, 1, 0, out.data_ptr(), chain, frac, ctypes.c_uint64(seed), None))

# the judge of every form: bsq_augment_device, then the generic token kernel
ref_c = pristine.clone()
capi.check(lib.bsq_augment_device(ref_c.data_ptr(), dof.data_ptr(), n, chain, frac, ctypes.c_uint64(7), None))
ref_t = torch.empty_like(out)
capi.check(lib.bsq_tokenize_device_generic(ctypes.byref(desc), ref_c.data_ptr(), dof.data_ptr(), n, P, 1, 0, ref_t.data_ptr(), None))
for v in vals:
    capi.check(lib.bsq_tuning_set(b"augment_fused", v))
    buf = pristine.clone()
    out.fill_(99)
    step(buf, 7)
    torch.cuda.synchronize()
    print("knob %d: chars equal %s  tokens equal %s" % (v, torch.equal(buf, ref_c), torch.equal(out, ref_t)), flush=True)
for rnd in range(3):
    row = []
    for v in vals:
        capi.check(lib.bsq_tuning_set(b"augment_fused", v))
        buf = pristine.clone()
        for i in range(300):
            step(buf, i)
        nsus = 6000
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for i in range(nsus):
            if i % 64 == 63:
                buf.copy_(pristine)
            step(buf, i)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / nsus * 1e3
        row.append("%d: %.2f us (frac %.3f)" % (v, us, algo / (us * 1e-6) / 8e12))
    print("  round %d  " % rnd + " | ".join(row), flush=True)
capi.check(lib.bsq_tuning_set(b"augment_fused", 0))
