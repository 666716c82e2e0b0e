#!/usr/bin/env python3
"""BSQ_PROFILE_HOST=1 python3 scripts/prof_host_phases.py [device|numpy8|numpyf|tokens]: host phases of the list -> result call on cfg3's batch
(scan, output allocation, pack, upload + launch [+ download]) and when the call returns / the GPU is done."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bioseq_amd as bsq
from bioseq_amd import synth
mode = sys.argv[1] if len(sys.argv) > 1 else "device"
c = synth.CONFIGS["cfg3"]
chars, offs = synth.synth_packed(c["seed"], c["n"], c["lo"], c["hi"], c["letters"])
seqs = synth.unpack(chars, offs)
tok = bsq.Tokenizer("AMINO20")
call = {"device": lambda: tok.batch_onehot_encode(seqs, padlen=1024, destchar="f", device="cuda"),
        "numpy8": lambda: tok.batch_onehot_encode(seqs, padlen=1024),
        "numpyf": lambda: tok.batch_onehot_encode(seqs, padlen=1024, destchar="f"),
        "tokens": lambda: tok.batch_tokenize(seqs, padlen=1024)}[mode]
for i in range(6):
    t0 = time.perf_counter(); r = call(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%s: call returned after %.2f ms, synced after %.2f ms" % (mode, (t1 - t0) * 1e3, (t2 - t0) * 1e3), file=sys.stderr)
    del r
