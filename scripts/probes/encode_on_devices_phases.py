"""Host phases of sharding.encode_on_devices on cfg3's list (65 536 sequences -> f32 one-hot), devices = N entries of cuda:0."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bioseq_amd
from bioseq_amd import synth, sharding, cbioseq
c = synth.CONFIGS["cfg3"]
chars, offs = synth.synth_packed(c["seed"], c["n"], c["lo"], c["hi"], c["letters"])
seqs = synth.unpack(chars, offs)
tok = bioseq_amd.Tokenizer(c["key"], bool(c["eos"]), bool(c["bos"]), bool(c["padchar"]))
P = c["padlen"]
def med(fn, n=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); r = fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t0)); del r
    a = np.median(np.array(ts), axis=0) * 1e3
    return "issue %.3f ms, done %.3f ms" % (a[0], a[1])
for nt in (8, 16):
    print("nthreads", nt)
    print("  scan only            ", med(lambda: cbioseq._ListScan(seqs, P, nt)))
    sc = cbioseq._ListScan(seqs, P, nt)
    buf = torch.empty(int(sc.offsets[-1]) + 16, dtype=torch.uint8, pin_memory=True).numpy()
    print("  pack all (pinned)    ", med(lambda: sc.pack(0, sc.n, buf)))
    for G in (1, 2, 4, 8):
        devs = ["cuda:0"] * G
        print("  encode_on_devices G=%d shards" % G, med(lambda: sharding.encode_on_devices(tok, seqs, P, "f", devices=devs, op="onehot", nthreads=nt)))
        print("  encode_on_devices G=%d root  " % G, med(lambda: sharding.encode_on_devices(tok, seqs, P, "f", devices=devs, op="onehot", nthreads=nt, root="cuda:0")))
    print("  list -> device (pieces)", med(lambda: tok.batch_onehot_encode(seqs, padlen=P, destchar="f", nthreads=nt, device="cuda")))
