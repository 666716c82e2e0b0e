"""One epoch of FlatFileDataset.batches() (device-side sampler: gather -> [augment] -> encode per batch, the store resident in HBM):
microseconds per batch, host time and GPU-complete time, for batch sizes a training loop uses.
    python3 scripts/loader_step_lab.py > gpurun_out/r04/loader_step_lab.txt"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bioseq_amd as bsq  # noqa: E402
from bioseq_amd import flatfile, loaders, synth  # noqa: E402

n = 200000
chars, offs = synth.synth_packed(3, n, 30, 512, synth.AA)
with tempfile.TemporaryDirectory() as d:
    fa = os.path.join(d, "x.fa")
    with open(fa, "w") as f:
        buf = chars.tobytes().decode()
        for i in range(n):
            f.write(">s%d\n%s\n" % (i, buf[offs[i]:offs[i + 1]]))
    ff = flatfile.FlatFile(fa, os.path.join(d, "x.ff"))
    tok = bsq.Tokenizer("SEB8", True, True, True)
    for label, kw in (("tokens int64", dict()), ("tokens int8 + BLOSUM62 augmentation", dict(augment=1, token_dtype="b")),
                      ("one-hot (B,C,P) f32", dict(cnn=True))):
        ds = loaders.FlatFileDataset(ff, tok, device="cuda", **kw)
        for bs in (64, 256, 1024, 4096):
            for _ in ds.batches(bs):  # warm-up epoch (allocator, staging)
                pass
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nb = 0
            for x in ds.batches(bs):
                nb += 1
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print(f"{label:38s} batch {bs:5d}: {nb:5d} batches, host {1e6 * (t1 - t0) / nb:7.1f} us per batch, "
                  f"GPU done after {1e6 * (t2 - t0) / nb:7.1f} us per batch", flush=True)
