import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bioseq_amd as bsq
from bioseq_amd import synth
c = synth.CONFIGS["cfg3"]
chars, offs = synth.synth_packed(c["seed"], c["n"], c["lo"], c["hi"], c["letters"])
seqs = synth.unpack(chars, offs)
tok = bsq.Tokenizer("AMINO20")
for i in range(8):
    t0 = time.perf_counter(); r = tok.batch_onehot_encode(seqs, padlen=1024, destchar="f", device="cuda"); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("call returned after %.2f ms, synced after %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3), file=sys.stderr)
    del r
