#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) timing of the Python list-of-bytes API on cfg3 / cfg2 -- never the bench `value`."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bioseq_amd as bsq
from bioseq_amd import synth
c = synth.CONFIGS["cfg3"]
chars, offs = synth.synth_packed(c["seed"], c["n"], c["lo"], c["hi"], c["letters"])
seqs = synth.unpack(chars, offs)
strs = synth.unpack(chars, offs, as_str=True)
tok = bsq.Tokenizer("AMINO20")
P = 1024
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0); del r
    return np.median(ts) * 1e3
print("host cpus", os.cpu_count())
for nt in (1, 8, 32):
    print("onehot list[bytes] -> device tensor, nthreads=%d : %.2f ms" % (nt, t(lambda: tok.batch_onehot_encode(seqs, padlen=P, destchar="f", nthreads=nt, device="cuda"))))
print("onehot list[str]   -> device tensor, nthreads=8 : %.2f ms" % t(lambda: tok.batch_onehot_encode(strs, padlen=P, destchar="f", nthreads=8, device="cuda")))
print("tokens list[bytes] -> device tensor (B,P)       : %.2f ms" % t(lambda: tok.batch_tokenize(seqs, padlen=P, batch_first=True, nthreads=8, device="cuda")))
print("tokens list[bytes] -> numpy (B,P) incl. D2H     : %.2f ms" % t(lambda: tok.batch_tokenize(seqs, padlen=P, batch_first=True, nthreads=8)))
print("onehot packed numpy -> device tensor            : %.2f ms" % t(lambda: tok.onehot_packed(chars, offs, P, "f", device="cuda")))
dch, dof = torch.from_numpy(chars).cuda(), torch.from_numpy(offs).cuda()
print("onehot packed device -> device tensor (alloc+kernel): %.2f ms" % t(lambda: tok.onehot_packed(dch, dof, P, "f")))
print("onehot packed device, validate=False            : %.2f ms" % t(lambda: tok.onehot_packed(dch, dof, P, "f", validate=False)))
def pipelined(fn, n=20):
    """n calls back to back, one sync at the end: what a training loop sees (host packing of batch i+1 overlaps the GPU work of batch i)."""
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn(); del r  # the caching allocator hands the same block to the next call (stream-ordered reuse)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for nt in (1, 8):
    print("onehot list[bytes] -> device tensor, nthreads=%d, 20 calls pipelined: %.2f ms per batch" % (nt, pipelined(lambda: tok.batch_onehot_encode(seqs, padlen=P, destchar="f", nthreads=nt, device="cuda"))))
print("tokens list[bytes] -> device tensor, nthreads=8, 20 calls pipelined: %.2f ms per batch" % pipelined(lambda: tok.batch_tokenize(seqs, padlen=P, batch_first=True, nthreads=8, device="cuda")))
from bioseq_amd import capi
lib = capi.load()
for nt in (1, 2, 4, 8, 16, 32):
    capi.check(lib.bsq_tuning_set(b"host_copy_threads", nt))
    for nseq in (8192, 65536):
        ms = t(lambda: tok.batch_onehot_encode(seqs[:nseq], padlen=P, destchar="f"), n=3)
        gb = P * nseq * 20 * 4 / 1e9
        print("onehot %5d seqs -> numpy (%.2f GB D2H), host_copy_threads=%2d: %7.1f ms = %5.1f GB/s" % (nseq, gb, nt, ms, gb / ms * 1e3), flush=True)
capi.check(lib.bsq_tuning_set(b"host_copy_threads", 0))
print("tokens list[bytes] -> numpy (B,P) incl. D2H     : %.2f ms" % t(lambda: tok.batch_tokenize(seqs, padlen=P, batch_first=True, nthreads=8)))
