"""Where a small call's microseconds go (64 and 1000 DNA sequences, tokens, batch_first)."""
import statistics
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bioseq_amd as bsq  # noqa: E402
from bioseq_amd import synth  # noqa: E402


def med_us(fn, n=300, sync=False):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        r = fn()
        if sync:
            torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        del r
    torch.cuda.synchronize()
    return statistics.median(ts) * 1e6


tok = bsq.Tokenizer("DNA", True, True, True)
for B in (64, 1000):
    chars, offs = synth.synth_packed(5, B, 1, 254, "ACGT")
    seqs = synth.unpack(chars, offs, as_str=True)
    dch, dof = torch.from_numpy(chars).cuda(), torch.from_numpy(offs).cuda()
    print(f"B={B}")
    print("  torch.empty((B,256), int8, cuda)            %6.1f us" % med_us(lambda: torch.empty((B, 256), dtype=torch.int8, device="cuda")))
    print("  torch.cuda.synchronize() on an idle device  %6.1f us" % med_us(lambda: torch.cuda.synchronize()))
    print("  tokenize_packed(device tensors), no sync    %6.1f us" % med_us(lambda: tok.tokenize_packed(dch, dof, 256, "b", True)))
    print("  tokenize_packed(device tensors, validate=False), no sync %6.1f us" % med_us(lambda: tok.tokenize_packed(dch, dof, 256, "b", True, validate=False)))
    print("  tokenize_packed(device tensors) + sync      %6.1f us" % med_us(lambda: tok.tokenize_packed(dch, dof, 256, "b", True), sync=True))
    print("  batch_tokenize(list, device), no sync       %6.1f us" % med_us(lambda: tok.batch_tokenize(seqs, padlen=256, batch_first=True, device="cuda")))
    print("  batch_tokenize(list, device) + sync         %6.1f us" % med_us(lambda: tok.batch_tokenize(seqs, padlen=256, batch_first=True, device="cuda"), sync=True))
    print("  batch_tokenize(list) -> numpy               %6.1f us" % med_us(lambda: tok.batch_tokenize(seqs, padlen=256, batch_first=True)))
    print("  numpy -> torch.from_numpy(...).cuda() + sync %5.1f us  (what the reference's callers add to get the batch onto the GPU)"
          % med_us(lambda: torch.from_numpy(np.zeros((B, 256), np.int8)).cuda(), sync=True))
