"""Where do the ~15 us per batch of `FlatFileDataset.batches(prefetch=k)` go?  Variants of one shuffled epoch (cfg5's store, 4096 per batch):
in order on the default stream / on a side stream; prefetch with and without record_stream; hand-off only."""
import os, sys, time, tempfile
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bioseq_amd
from bioseq_amd import synth
from bioseq_amd.flatfile import FlatFile
from bioseq_amd.loaders import FlatFileDataset

dev = torch.device("cuda:0")
c = synth.CONFIGS["cfg5"]
chars, offs = synth.synth_packed(c["seed"], c["n"], c["lo"], c["hi"], c["letters"])
tmp = tempfile.mkdtemp()
path = os.path.join(tmp, "s.ff")
with open(path, "wb") as f:
    f.write(np.array([c["n"]], dtype="<u8").tobytes()); f.write(offs.astype("<u8").tobytes()); f.write(chars.tobytes())
ff = FlatFile(path)
tok = bioseq_amd.Tokenizer(c["key"], bool(c["eos"]), bool(c["bos"]), bool(c["padchar"]))
ds = FlatFileDataset(ff, tok, device=dev, token_dtype="b")
BS = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nb = -(-c["n"] // BS)


def epoch(fn):
    ts = []
    for ep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn(ep)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return np.median(ts[1:]) / nb * 1e6


def inorder(ep, pf=0):
    g = torch.Generator(device=dev).manual_seed(ep)
    for b in ds.batches(BS, generator=g, prefetch=pf):
        pass


side = torch.cuda.Stream()
def inorder_side(ep):
    with torch.cuda.stream(side):
        inorder(ep)

print("in order, default stream      %.1f us/batch" % epoch(inorder))
print("in order, one side stream     %.1f" % epoch(inorder_side))
print("prefetch 2                    %.1f" % epoch(lambda ep: inorder(ep, 2)))
real = torch.Tensor.record_stream
torch.Tensor.record_stream = lambda self, s: None
print("prefetch 2, no record_stream  %.1f" % epoch(lambda ep: inorder(ep, 2)))
torch.Tensor.record_stream = real
# the pieces of a batch step on the default stream
order = torch.randperm(c["n"], device=dev)
def gather_only(ep):
    for f in range(0, c["n"], BS):
        ds._packed_device(0, 0, order[f:f + BS], trusted=True)
print("gather only                   %.1f" % epoch(gather_only))
packs = [ds._packed_device(0, 0, order[f:f + BS], trusted=True) for f in range(0, c["n"], BS)]
def encode_only(ep):
    for ch, of in packs:
        ds._encode(ch, of)
print("encode only                   %.1f" % epoch(encode_only))
def slice_only(ep):
    for f in range(0, c["n"], BS):
        order[f:f + BS]
print("index slice only              %.1f" % epoch(slice_only))
ev = torch.cuda.Event()
def handoff_only(ep):
    cur = torch.cuda.current_stream()
    for f in range(0, c["n"], BS):
        back = torch.cuda.current_stream(); torch.cuda.set_stream(side); ev.record(side); torch.cuda.set_stream(back); cur.wait_event(ev)
print("stream switch + event + wait  %.1f" % epoch(handoff_only))
