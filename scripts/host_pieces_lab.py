"""list[bytes] -> seq-first one-hot tensor on the device (cfg3 batch), synchronous and back to back, for every value of the
host_pieces knob (1 = one upload + one encode, N = N pieces, 0 = automatic).  Prints medians; run on the GPU box:
    python3 scripts/host_pieces_lab.py > gpurun_out/r04/host_pieces_lab.txt"""
import statistics
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bioseq_amd as bsq  # noqa: E402
from bioseq_amd import capi, synth  # noqa: E402

lib = capi.load()
B, P = 65536, 1024
chars, offs = synth.synth_packed(1, B, 50, 1024, synth.AA)
items = [bytes(chars[offs[i]:offs[i + 1]]) for i in range(B)]
tok = bsq.Tokenizer("AMINO20")


def sync_ms(dest, n=15):
    ts = []
    r = None
    for _ in range(n):
        del r
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = tok.batch_onehot_encode(items, padlen=P, destchar=dest, device="cuda")
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(ts), min(ts), r


def pipelined_ms(dest, n=20):
    best = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            r = tok.batch_onehot_encode(items, padlen=P, destchar=dest, device="cuda")
            del r
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) * 1e3 / n)
    return min(best)


ref = {}
for dest in ("f", "b"):
    for knob in (1, 2, 4, 8, 0, 1, 4, 0):
        capi.check(lib.bsq_tuning_set(b"host_pieces", knob))
        for _ in range(3):
            tok.batch_onehot_encode(items, padlen=P, destchar=dest, device="cuda")
        med, lo, r = sync_ms(dest)
        fold = int(r.view(torch.uint8).to(torch.int64).sum().item())
        ref.setdefault(dest, fold)
        assert fold == ref[dest], "results differ between piece counts"
        del r
        pipe = pipelined_ms(dest)
        print(f"destchar={dest} host_pieces={knob}: synchronous median {med:.3f} ms (min {lo:.3f}), back to back x20 {pipe:.3f} ms/batch", flush=True)
capi.check(lib.bsq_tuning_set(b"host_pieces", 0))
