import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, bioseq_amd as bsq
from bioseq_amd import synth
c = synth.CONFIGS["cfg3"]
chars, offs = synth.synth_packed(c["seed"], c["n"], c["lo"], c["hi"], c["letters"])
seqs = synth.unpack(chars, offs)
tok = bsq.Tokenizer("AMINO20")
for nt in (1, 8):
    print("nthreads", nt, file=sys.stderr)
    for _ in range(4):
        r = tok.batch_onehot_encode(seqs, padlen=1024, destchar="f", nthreads=nt, device="cuda"); del r
    torch.cuda.synchronize()
