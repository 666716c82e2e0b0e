"""rocprofv3 --kernel-trace target: the list -> device one-hot of cfg3 (65 536 sequences, 5.37 GB f32) through the Python surface, which uploads and
encodes the batch in PIECES (Tokenizer::staged): five calls.  The trace's kernel start / end timestamps show what separates the pieces' kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bioseq_amd
from bioseq_amd import synth
c = synth.CONFIGS["cfg3"]
chars, offs = synth.synth_packed(c["seed"], c["n"], c["lo"], c["hi"], c["letters"])
seqs = synth.unpack(chars, offs)
tok = bioseq_amd.Tokenizer(c["key"], bool(c["eos"]), bool(c["bos"]), bool(c["padchar"]))
for _ in range(5):
    r = tok.batch_onehot_encode(seqs, padlen=c["padlen"], destchar="f", nthreads=8, device="cuda")
    torch.cuda.synchronize()
    del r
