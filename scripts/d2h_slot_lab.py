import statistics, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bioseq_amd as bsq
from bioseq_amd import synth
B, P = 65536, 1024
chars, offs = synth.synth_packed(1, B, 50, 1024, synth.AA)
items = [bytes(chars[offs[i]:offs[i + 1]]) for i in range(B)]
tok = bsq.Tokenizer("AMINO20")
def med(fn, n=7):
    r = fn(); del r
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); r = fn(); ts.append((time.perf_counter() - t0) * 1e3); del r
    return statistics.median(ts)
print("list -> numpy int8 one-hot (1.34 GB) %.2f ms" % med(lambda: tok.batch_onehot_encode(items, padlen=P)))
print("list -> numpy f32 one-hot (5.4 GB)  %.2f ms" % med(lambda: tok.batch_onehot_encode(items, padlen=P, destchar="f"), 4))
