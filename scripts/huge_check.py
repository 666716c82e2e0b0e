#!/usr/bin/env python3
"""A 43 GB one-hot (262144 x 1024 AMINO20 float64, both layouts) checked on the device through size-independent
properties: argmax == tokens inside the sequences, exactly one 1 per residue position, none behind them, ones == residues."""
import sys, os, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bioseq_amd as bsq
from bioseq_amd import synth
dev = torch.device("cuda:0")
B, P = 262144, 1024
chars, offs = synth.synth_packed(77, B, 50, 1024, synth.AA)
dch, dof = torch.from_numpy(chars).to(dev), torch.from_numpy(offs).to(dev)
tok = bsq.Tokenizer("AMINO20")
for d, lay in (("d", "tbc"), ("d", "bcl"), ("l", "tok")):
    if lay == "tok":
        out = tok.tokenize_packed(dch, dof, P, d, True)   # (B,P) int64: 2.1 GB
        ref = tok.tokenize_packed(dch, dof, P, "B", True)
        assert torch.equal(out.to(torch.int8), ref); print("tokens int64 ok", out.numel() * 8 / 1e9, "GB"); continue
    out = tok.onehot_packed(dch, dof, P, d, layout=lay)
    print(lay, tuple(out.shape), out.numel() * 8 / 1e9, "GB")
    tokens = tok.tokenize_packed(dch, dof, P, "B", False)  # (P,B) int8; unmapped never occurs here, no pad -> zeros beyond L
    lens = torch.from_numpy(np.diff(offs)).to(dev)
    total = 0
    step = 64
    for p0 in range(0, P, step):
        sl = out[p0:p0 + step] if lay == "tbc" else out[:, :, p0:p0 + step].permute(2, 0, 1)
        am = sl.argmax(-1).to(torch.int8)
        valid = (torch.arange(p0, p0 + step, device=dev)[:, None] < lens[None, :])
        assert torch.equal(torch.where(valid, am, torch.zeros_like(am)), torch.where(valid, tokens[p0:p0 + step], torch.zeros_like(am))), p0
        s = sl.sum(-1)
        assert torch.equal(s == 1, valid), p0
        total += int(s.sum().item())
    assert total == int(offs[-1]), (total, int(offs[-1]))
    print(lay, "ok: ones == residues =", total)
    del out
