import statistics, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bioseq_amd as bsq
from bioseq_amd import synth
B, P = 65536, 1024
chars, offs = synth.synth_packed(1, B, 50, 1024, synth.AA)
tok = bsq.Tokenizer("AMINO20")
def med(fn, n=12):
    r = fn(); del r
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3); del r
    return statistics.median(ts)
print("packed numpy -> f32 one-hot on the device + sync   %.3f ms" % med(lambda: tok.onehot_packed(chars, offs, P, "f", device="cuda")))
print("packed numpy -> int8 (B,P) tokens on the device    %.3f ms" % med(lambda: tok.tokenize_packed(chars, offs, P, "b", True, device="cuda")))
print("packed numpy -> int8 (P,B) tokens -> numpy         %.3f ms" % med(lambda: tok.tokenize_packed(chars, offs, P, "b", False)))
