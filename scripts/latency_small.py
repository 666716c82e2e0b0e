#!/usr/bin/env python3
"""Per-call latency of the Python API on small batches (dataloader-sized)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bioseq_amd as bsq
from bioseq_amd import synth
tok = bsq.Tokenizer("AMINO20", 1, 1, 1)
def t(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for B, P in ((1, 512), (32, 512), (256, 512), (1024, 1024)):
    seqs = synth.unpack(*synth.synth_packed(B, B, 50, P - 2, synth.AA))
    chars, offs = synth.synth_packed(B, B, 50, P - 2, synth.AA)
    dch, dof = torch.from_numpy(chars).cuda(), torch.from_numpy(offs).cuda()
    print("B=%4d P=%4d | tokens list->device %.0f us | onehot list->device %.0f us | tokens list->numpy %.0f us | packed-device tokens %.0f us (validate=False %.0f us) | packed-device onehot %.0f us" % (
        B, P, t(lambda: tok.batch_tokenize(seqs, padlen=P, batch_first=True, device="cuda")),
        t(lambda: tok.batch_onehot_encode(seqs, padlen=P, destchar="f", device="cuda")),
        t(lambda: tok.batch_tokenize(seqs, padlen=P, batch_first=True)),
        t(lambda: tok.tokenize_packed(dch, dof, P, "B", True)), t(lambda: tok.tokenize_packed(dch, dof, P, "B", True, validate=False)),
        t(lambda: tok.onehot_packed(dch, dof, P, "f", validate=False))))
