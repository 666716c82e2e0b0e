#!/usr/bin/env python3
"""Shapes far from the BASELINE configs -- very long padlens, very short reads in millions, the BYTES alphabet -- through the packed device
entries: microseconds, GB/s of algorithmic bytes, fraction of 8 TB/s; every result checked against the ORACLE on a sample of sequences
(first / last 64) when the whole tensor is too large for the host.      extreme_shapes.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioseq_amd import capi, synth
from oracle import oracle as O
lib = capi.load()
dev = torch.device("cuda:0")
VALIDATE = os.environ.get("VALIDATE", "0") == "1"   # (the default of the packed calls is to validate the lengths on the device first: a launch + a host read)
SHAPES = [("onehot", "AMINO20", (0, 0, 0), 16384, 100, 4096, 4096, "f", None), ("onehot", "AMINO20", (1, 1, 1), 4096, 1000, 16382, 16384, "f", None),
          ("onehot", "DNA4", (1, 1, 1), 4000000, 20, 30, 32, "f", None), ("onehot", "DNA4", (1, 1, 1), 4000000, 20, 30, 32, "B", None),
          ("tokens", "AMINO20", (0, 0, 0), 16384, 1000, 16384, 16384, "b", True), ("tokens", "AMINO20", (0, 0, 0), 16384, 1000, 16384, 16384, "b", False),
          ("tokens", "DNA4", (1, 1, 1), 4000000, 150, 150, 160, "b", True), ("tokens", "DNA4", (1, 1, 1), 4000000, 150, 150, 160, "b", False),
          ("onehot", "BYTES", (0, 0, 0), 4096, 50, 256, 256, "B", None), ("onehot", "BYTES", (0, 0, 0), 4096, 50, 256, 256, "f", None),
          ("onehot", "AMINO20", (0, 0, 0), 1000000, 100, 160, 160, "f", None), ("onehot", "AMINO20", (0, 0, 0), 2000000, 100, 160, 160, "B", None)]
for si, (op, key, flags, B, lo, hi, P, dc, bf) in enumerate(SHAPES):
    letters = "ACGT" if key == "DNA4" else synth.AA
    chars, offs = synth.synth_packed(9000 + si, B, lo, hi, letters)
    tok = __import__("bioseq_amd").Tokenizer(key, *flags)
    ora = O.OracleTokenizer(key, *flags)
    dch, dof = torch.from_numpy(chars).to(dev), torch.from_numpy(offs).to(dev)
    if op == "onehot":
        f = lambda: tok.onehot_packed(dch, dof, P, dc, validate=VALIDATE)
    else:
        f = lambda: tok.tokenize_packed(dch, dof, P, dc, bf, validate=VALIDATE)
    out = f(); torch.cuda.synchronize()
    # oracle on the first and the last 64 sequences
    ok = True
    for b0 in (0, B - 64):
        o = offs[b0:b0 + 65]; c = chars[int(o[0]):int(o[-1])]; oo = (o - o[0]).copy()
        want = ora.onehot_packed(c, oo, P, dc) if op == "onehot" else ora.tokenize_packed(c, oo, P, dc, bf)
        got = (out[:, b0:b0 + 64] if (op == "onehot" or not bf) else out[b0:b0 + 64]).contiguous().cpu().numpy()
        ok = ok and got.tobytes() == np.ascontiguousarray(want).tobytes()
    ob = out.numel() * out.element_size()
    algo = int(offs[-1]) + 8 * (B + 1) + ob
    del out
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r = f(); del r
        a.record()
        for _ in range(5):
            r = f(); del r
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 5)
    t = float(np.median(ts))
    print("%-6s %-8s %s B=%8d P=%6d %s%s out=%6.2f GB | %9.1f us incl. result allocation%s  %6.0f GB/s  frac %.3f  %s" % (
        op, key, flags, B, P, dc, "" if bf is None else (" (B,P)" if bf else " (P,B)"), ob / 1e9, t * 1e3, " + validation" if VALIDATE else "", algo / t / 1e6, algo / t / 8e9, "ok" if ok else "MISMATCH"), flush=True)
    del dch, dof
    torch.cuda.empty_cache()
