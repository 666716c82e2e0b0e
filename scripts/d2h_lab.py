#!/usr/bin/env python3
"""Round 3: list -> numpy (the reference's default return) on cfg3 -- the 5.37 GB result through the pinned ring, with and
without transparent huge pages for the destination array (knob host_hugepages) and by worker threads."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bioseq_amd as bsq
from bioseq_amd import capi, synth
lib = capi.load()
c = synth.CONFIGS["cfg3"]
chars, offs = synth.synth_packed(c["seed"], c["n"], c["lo"], c["hi"], c["letters"])
seqs = synth.unpack(chars, offs)
tok = bsq.Tokenizer("AMINO20")
print("THP:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
def med(fn, n=5):
    r = fn(); del r; ts = []
    for _ in range(n):
        t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0); del r
    return np.median(ts) * 1e3
for rnd in range(2):
    for hp in (1, 0):
        for th in (8, 16):
            capi.check(lib.bsq_tuning_set(b"host_hugepages", hp)); capi.check(lib.bsq_tuning_set(b"host_copy_threads", th))
            ms = med(lambda: tok.batch_onehot_encode(seqs, padlen=1024, destchar="f"))
            print("  hugepages %s, %2d copy threads: list -> numpy %.1f ms = %.1f GB/s" % ("off" if hp else "on ", th, ms, 5.37e3 / ms), flush=True)
capi.check(lib.bsq_tuning_set(b"host_hugepages", 0)); capi.check(lib.bsq_tuning_set(b"host_copy_threads", 0))
