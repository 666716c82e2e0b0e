"""Small batches through the drop-in surface: BASELINE config 1 (1000 DNA sequences, len <= 256, padlen 256, batch_first) and its
neighbours -- latency per call of the product (numpy result as the reference returns it; device result + synchronize) beside the
reference's own C++ on the box's CPU (oracle/_ref, nthreads = 1 as the reference defaults, and all threads).
    python3 scripts/small_batch_lab.py > gpurun_out/r04/small_batch_lab.txt"""
import statistics
import sys
import time

import torch

sys.path.insert(0, ".")
import bioseq_amd as bsq  # noqa: E402
from bioseq_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402  (baseline only)

ref = O.load_reference()


def med_us(fn, n=200, sync=False):
    for _ in range(10):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        r = fn()
        if sync:
            torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        del r
    return statistics.median(ts) * 1e6


for B, hi, P in ((1000, 256, 256), (64, 256, 256), (4096, 256, 256), (1000, 1024, 1024), (16384, 256, 256)):
    chars, offs = synth.synth_packed(5, B, 1, hi - 2, "ACGT")
    seqs = synth.unpack(chars, offs, as_str=True)
    tok = bsq.Tokenizer("DNA", True, True, True)
    row = [f"B={B:6d} len<={hi:4d} padlen={P:4d}"]
    for name, fn, sync in (("tokens -> numpy", lambda: tok.batch_tokenize(seqs, padlen=P, batch_first=True), False),
                           ("tokens -> device+sync", lambda: tok.batch_tokenize(seqs, padlen=P, batch_first=True, device="cuda"), True),
                           ("one-hot f32 -> device+sync", lambda: tok.batch_onehot_encode(seqs, padlen=P, destchar="f", device="cuda"), True),
                           ("one-hot int8 -> numpy", lambda: tok.batch_onehot_encode(seqs, padlen=P), False)):
        row.append(f"{name} {med_us(fn, sync=sync):8.1f} us")
    if ref is not None:
        rt = ref.Tokenizer("DNA", True, True, True)
        row.append(f"| reference C++ tokens nthreads=1 {med_us(lambda: rt.batch_tokenize(seqs, padlen=P, batch_first=True, nthreads=1), 50):8.1f} us")
        row.append(f"one-hot int8 nthreads=1 {med_us(lambda: rt.batch_onehot_encode(seqs, padlen=P, nthreads=1), 20):9.1f} us")
        row.append(f"one-hot int8 nthreads=32 {med_us(lambda: rt.batch_onehot_encode(seqs, padlen=P, nthreads=32), 20):9.1f} us")
    print("  ".join(row), flush=True)
