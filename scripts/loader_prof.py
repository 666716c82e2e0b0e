import cProfile, pstats, os, sys, tempfile
import torch
sys.path.insert(0, ".")
import bioseq_amd as bsq
from bioseq_amd import flatfile, loaders, synth
n = 100000
chars, offs = synth.synth_packed(3, n, 30, 512, synth.AA)
seqs = synth.unpack(chars, offs)
with tempfile.TemporaryDirectory() as d:
    ff = flatfile.FlatFile(flatfile.write_flatfile(seqs, os.path.join(d, "x.ff")))
    tok = bsq.Tokenizer("SEB8", True, True, True)
    ds = loaders.FlatFileDataset(ff, tok, device="cuda")
    for _ in ds.batches(256): pass
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in ds.batches(256): pass
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
