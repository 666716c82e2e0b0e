#!/usr/bin/env python3
"""k_tokens_bp8 lab: correctness of every variant against k_tokenize_chunks / the generic kernel, then timings
(variants, occupancy caps, ablations) on the cfg2 / cfg5 batches.  Run on the GPU box."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioseq_amd import capi, synth
lib = capi.load()
dev = torch.device("cuda:0")
def setk(**kw):
    for k, v in kw.items(): capi.check(lib.bsq_tuning_set(k.encode(), v))
def tok(desc, dch, dof, B, P, out):
    capi.check(lib.bsq_tokenize_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), B, P, 1, 0, out.data_ptr(), None))
def gen(desc, dch, dof, B, P, out):
    capi.check(lib.bsq_tokenize_device_generic(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), B, P, 1, 0, out.data_ptr(), None))

VARIANTS = [dict(tokens8_lookup=2, tokens8_nch=1), dict(tokens8_lookup=1, tokens8_nch=1),
            dict(tokens8_lookup=2, tokens8_nch=2), dict(tokens8_lookup=1, tokens8_nch=3)]
ok = True
shapes = [("AMINO20", (0, 0, 0), 3000, 0, 300, 304, synth.AA), ("DNA", (1, 1, 1), 5000, 1, 254, 256, "ACGT"),
          ("SEB8", (1, 0, 1), 777, 0, 510, 512, synth.DIRTY), ("DNA5", (0, 1, 0), 1234, 100, 127, 128, synth.DIRTY),
          ("AMINO20", (1, 1, 1), 65, 0, 4100, 4112, synth.DIRTY), ("AMINO20", (1, 1, 0), 1, 0, 0, 128, synth.AA),
          ("AMINO20", (0, 0, 1), 3, 5, 9, 16 * 9, synth.DIRTY), ("DAYHOFF", (1, 1, 1), 40000, 0, 126, 128, synth.DIRTY)]
for key, flags, B, lo, hi, P, letters in shapes:
    chars, offs = synth.synth_packed(42 + B, B, lo, hi, letters)
    if len(chars) == 0: chars = np.zeros(0, np.uint8)
    desc = capi.make_desc(key, *flags)
    dch = torch.from_numpy(np.concatenate([chars, np.zeros(1, np.uint8)])).to(dev)[:len(chars)]
    dof = torch.from_numpy(offs).to(dev)
    ref = torch.empty((B, P), dtype=torch.int8, device=dev); gen(desc, dch, dof, B, P, ref)
    for v in VARIANTS:
        setk(tokens8=0, **v)
        out = torch.full((B, P), 99, dtype=torch.int8, device=dev); tok(desc, dch, dof, B, P, out)
        torch.cuda.synchronize()
        same = bool(torch.equal(out, ref))
        ok &= same
        if not same:
            bad = (out != ref).nonzero()
            print("MISMATCH", key, flags, B, P, v, "first", bad[:5].tolist(), out[bad[0][0], bad[0][1]].item(), ref[bad[0][0], bad[0][1]].item())
print("correctness:", "OK" if ok else "FAILED", flush=True)

def timeit(fn, n=40, warm=15):
    for _ in range(warm): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) * 1e3 for a, b in ev]
    return np.median(ts), np.min(ts)

for cfgname in ("cfg2", "cfg5"):
    cfg = synth.CONFIGS[cfgname]
    B, P = cfg["n"], cfg["padlen"]
    chars, offs = synth.synth_packed(cfg["seed"], B, cfg["lo"], cfg["hi"], cfg["letters"])
    desc = capi.make_desc(cfg["key"], cfg["eos"], cfg["bos"], cfg["padchar"])
    dch, dof = torch.from_numpy(chars).to(dev), torch.from_numpy(offs).to(dev)
    out = torch.empty((B, P), dtype=torch.int8, device=dev)
    ref = torch.empty((B, P), dtype=torch.int8, device=dev)
    setk(tokens8=1); tok(desc, dch, dof, B, P, ref)
    algo = len(chars) + 8 * (B + 1) + B * P
    fn = lambda: tok(desc, dch, dof, B, P, out)
    def report(tag):
        med, mn = timeit(fn)
        print("  %-44s median %6.2f us  min %6.2f  -> %5.0f GB/s (%.3f of 8 TB/s)" % (tag, med, mn, algo / med / 1e3, algo / med / 1e3 / 8000), flush=True)
    print(cfgname, "B", B, "P", P, "algorithmic bytes", algo)
    for rnd in range(1):
        setk(tokens8=1, tokens8_abl=0, tokens8_pad=0); report("k_tokenize_chunks (round 1 kernel)")
        for v in VARIANTS:
            setk(tokens8=0, **v)
            out.fill_(99); fn(); torch.cuda.synchronize()
            report("k_tokens_bp8 %s %s" % (v, "" if torch.equal(out, ref) else "MISMATCH"))
    setk(tokens8=0, tokens8_lookup=2, tokens8_nch=1, tokens8_pad=0)
    def check(tag):
        out.fill_(99); fn(); torch.cuda.synchronize()
        if not torch.equal(out, ref): print("   MISMATCH in", tag)
    for rnd in range(2):
        for sk in (0, 1, 2):
            for lk in (0, 1, 2, 3, 4, 5):
                setk(tokens8_store=sk, tokens8_load=lk); check("load kind"); report("store kind %d load kind %d" % (sk, lk))
    setk(tokens8_store=0, tokens8_load=0)
    capi.check(lib.bsq_tuning_set(b"fill_mode", 1))
    fn = lambda: capi.check(lib.bsq_fill_device(out.data_ptr(), B * P, 0, None))
    med, mn = timeit(fn); print("  plain fill of the output: median %.2f us min %.2f -> %.0f GB/s" % (med, mn, B * P / med / 1e3))
