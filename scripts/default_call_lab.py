"""The reference's literal default call tok.batch_tokenize(seqs, padlen) -> numpy int8 (P, B) on the cfg2 batch: the whole call against
its parts (list -> device tensor in pieces; device tensor -> numpy through torch)."""
import statistics
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bioseq_amd as bsq  # noqa: E402
from bioseq_amd import synth  # noqa: E402

B, P = 65536, 1024
chars, offs = synth.synth_packed(1, B, 50, 1024, synth.AA)
items = [bytes(chars[offs[i]:offs[i + 1]]) for i in range(B)]
tok = bsq.Tokenizer("AMINO20")


def med(fn, n=12):
    r = fn()
    del r
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        del r
    return statistics.median(ts)


for bf in (False, True):
    lay = "(B,P)" if bf else "(P,B)"
    print(f"{lay} list -> numpy (the default call)         {med(lambda: tok.batch_tokenize(items, padlen=P, batch_first=bf)):.3f} ms")
    print(f"{lay} list -> device tensor + sync            {med(lambda: tok.batch_tokenize(items, padlen=P, batch_first=bf, device='cuda')):.3f} ms")
    d = tok.batch_tokenize(items, padlen=P, batch_first=bf, device="cuda")
    print(f"{lay} device tensor -> .cpu().numpy()         {med(lambda: d.cpu().numpy()):.3f} ms")
    pin = torch.empty(d.shape, dtype=d.dtype, pin_memory=True)
    print(f"{lay} device tensor -> pinned (copy_)         {med(lambda: pin.copy_(d, non_blocking=True)):.3f} ms")
    print(f"{lay} pinned -> fresh numpy (np.array copy)   {med(lambda: np.array(pin.numpy(), copy=True)):.3f} ms")
    print(f"{lay} list -> device, then .cpu().numpy()     {med(lambda: tok.batch_tokenize(items, padlen=P, batch_first=bf, device='cuda').cpu().numpy()):.3f} ms")
