import ctypes, os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
from bioseq_amd import capi, synth, blosum
import bioseq_amd as bsq
lib = capi.load(); dev = torch.device("cuda:0")
tok = bsq.Tokenizer("SEB8")
for B in (256, 1024, 4096, 16384, 65536, 262144):
    P = 512
    chars, offs = synth.synth_packed(5, B, 30, 510, synth.AA)
    dch0, dof = torch.from_numpy(chars).to(dev), torch.from_numpy(offs).to(dev)
    out = torch.empty((B, P), dtype=torch.int8, device=dev)
    res = []
    for knob in (1, 0):
        capi.check(lib.bsq_tuning_set(b"augment_fused", knob))
        dch = dch0.clone()
        def run(i): blosum.augment_tokenize_packed(tok, dch, dof, P, "b", True, chain_len=1, augment_frac=0.5, seed=i, out=out)
        for i in range(30): run(i)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(200):
            if i % 32 == 31: dch.copy_(dch0)
            run(100 + i)
        b.record(); torch.cuda.synchronize()
        res.append(a.elapsed_time(b) * 5)
    print("B=%6d P=512: two launches %6.1f us | one launch %6.1f us" % (B, res[0], res[1]), flush=True)
capi.check(lib.bsq_tuning_set(b"augment_fused", 0))
torch.cuda.synchronize()
print("fused status:", lib.bsq_fused_status(None))
