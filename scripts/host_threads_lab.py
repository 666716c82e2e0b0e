"""list[bytes] -> seq-first one-hot on the device (cfg3 batch, f32): synchronous median per nthreads (0 = automatic)."""
import statistics
import sys
import time

import torch

sys.path.insert(0, ".")
import bioseq_amd as bsq  # noqa: E402
from bioseq_amd import synth  # noqa: E402

B, P = 65536, 1024
chars, offs = synth.synth_packed(1, B, 50, 1024, synth.AA)
items = [bytes(chars[offs[i]:offs[i + 1]]) for i in range(B)]
tok = bsq.Tokenizer("AMINO20")
for rep in range(2):
    for nt in (0, 4, 8, 16, 24, 32, 48, 64):
        ts = []
        for i in range(18):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = tok.batch_onehot_encode(items, padlen=P, destchar="f", device="cuda", nthreads=nt)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
            del r
        print(f"nthreads={nt}: synchronous median {statistics.median(ts[3:]):.3f} ms (min {min(ts):.3f})", flush=True)
