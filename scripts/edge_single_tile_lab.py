import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, bioseq_amd
from bioseq_amd import capi, synth
lib = capi.load(); dev = torch.device("cuda:0")
B, P = 16000000, 32
chars, offs = synth.synth_packed(5, B, 20, 30, "ACGT")
tok = bioseq_amd.Tokenizer("DNA4", 1, 1, 1)
dch, dof = torch.from_numpy(chars).to(dev), torch.from_numpy(offs).to(dev)
ref = None
for mb in (-1, 0):
    capi.check(lib.bsq_tuning_set(b"two_pass_slice_mb", mb))
    out = tok.onehot_packed(dch, dof, P, "f", validate=False); torch.cuda.synchronize()
    if ref is None: ref = out
    else: assert torch.equal(ref, out), "sequence blocks differ"
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            r = tok.onehot_packed(dch, dof, P, "f", validate=False); del r
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 3)
    ob = out.numel() * 4
    algo = int(offs[-1]) + 8 * (B + 1) + ob
    print("two_pass_slice_mb=%d: %.1f us  frac %.3f  (out %.1f GB)" % (mb, np.median(ts) * 1e3, algo / np.median(ts) / 8e9, ob / 1e9), flush=True)
    if mb == 0: del out
