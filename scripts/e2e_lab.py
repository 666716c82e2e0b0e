#!/usr/bin/env python3
"""Host-path lab (round 3): list[bytes] -> device tensor on cfg3, by thread count; BSQ_PACK_PIECE_LOG2 selects the upload piece."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bioseq_amd as bsq
from bioseq_amd import synth
c = synth.CONFIGS["cfg3"]
chars, offs = synth.synth_packed(c["seed"], c["n"], c["lo"], c["hi"], c["letters"])
seqs = synth.unpack(chars, offs)
tok = bsq.Tokenizer("AMINO20")
def med(fn, n=15):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0); del r
    return np.median(ts) * 1e3, np.min(ts) * 1e3
def pipe(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        r = fn(); del r
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("BSQ_PACK_PIECE_LOG2 =", os.environ.get("BSQ_PACK_PIECE_LOG2", "(default 20)"))
for nt in (0, 1, 4, 8, 16, 32):
    f = lambda: tok.batch_onehot_encode(seqs, padlen=1024, destchar="f", nthreads=nt, device="cuda")
    m, mn = med(f)
    print("  onehot list[bytes] -> device, nthreads=%2d: median %.2f ms  min %.2f ms   pipelined x20: %.2f ms/batch" % (nt, m, mn, pipe(f)), flush=True)
f = lambda: tok.batch_tokenize(seqs, padlen=1024, batch_first=True, device="cuda")
print("  tokens list[bytes] -> device, nthreads auto: median %.2f ms, pipelined %.2f" % (med(f)[0], pipe(f)))
