#!/usr/bin/env python3
"""Read + write yardstick lab (round 3): is k_tokens_bp8 on cfg2 / cfg5 at the rate of a plain copy of its shape?
Times, interleaved in one process: the real kernel, a fill of the output alone, bsq_copy_mix_device (the kernel's stream
shape with none of its work) with one / two dependent load steps and with stores that do not wait -- with the source
resident in the Infinity Cache (the same 35 / 67 MB every launch, as in bench.py) and cycling over > 256 MiB of sources."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioseq_amd import capi, synth
lib = capi.load()
dev = torch.device("cuda:0")
def setk(**kw):
    for k, v in kw.items(): capi.check(lib.bsq_tuning_set(k.encode(), v))

def loop_us(fn, n=60, warm=20, reps=5):
    for _ in range(warm): fn()
    res = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        res.append(a.elapsed_time(b) * 1e3 / n)
    return min(res), float(np.median(res))

for cfgname in sys.argv[1:] or ("cfg2", "cfg5"):
    cfg = synth.CONFIGS[cfgname]
    B, P = cfg["n"], cfg["padlen"]
    chars, offs = synth.synth_packed(cfg["seed"], B, cfg["lo"], cfg["hi"], cfg["letters"])
    desc = capi.make_desc(cfg["key"], cfg["eos"], cfg["bos"], cfg["padchar"])
    ncopies = max(2, int(300e6 // len(chars)) + 1)
    dchs = [torch.from_numpy(chars).to(dev) for _ in range(ncopies)]
    dof = torch.from_numpy(offs).to(dev)
    outs = [torch.empty((B, P), dtype=torch.int8, device=dev) for _ in range(5)]
    out = outs[0]
    nsrc = (len(chars) // 16) * 16
    algo = len(chars) + 8 * (B + 1) + B * P
    print("%s: B %d P %d, out %.1f MB, chars %.1f MB (%d copies for the cold runs), algorithmic %d B" % (cfgname, B, P, B * P / 1e6, len(chars) / 1e6, ncopies, algo), flush=True)
    state = {"i": 0}
    def kern(): capi.check(lib.bsq_tokenize_device(ctypes.byref(desc), dchs[0].data_ptr(), dof.data_ptr(), B, P, 1, 0, out.data_ptr(), None))
    def kern_cold():
        state["i"] = (state["i"] + 1) % ncopies
        capi.check(lib.bsq_tokenize_device(ctypes.byref(desc), dchs[state["i"]].data_ptr(), dof.data_ptr(), B, P, 1, 0, outs[state["i"] % 5].data_ptr(), None))
    def fill(): capi.check(lib.bsq_fill_device(out.data_ptr(), B * P, 7, None))
    def mix(mode, nt=1, cold=False):
        def f():
            if cold: state["i"] = (state["i"] + 1) % ncopies
            capi.check(lib.bsq_copy_mix_device(outs[state["i"] % 5 if cold else 0].data_ptr(), B * P, dchs[state["i"] if cold else 0].data_ptr(), nsrc, mode, nt, None))
        return f
    rows = [("k_tokens_bp8 (source resident)", kern, algo), ("k_tokens_bp8 (sources cycling, > 256 MiB)", kern_cold, algo)]
    for fm in (1, 3):
        rows.append(("fill of the output, fill_mode %d" % fm, ("fill", fm), B * P))
    for mode in (0, 1, 2):
        rows.append(("copy mix mode %d nt (resident)" % mode, mix(mode), nsrc + B * P))
    rows.append(("copy mix mode 0 plain stores (resident)", mix(0, 0), nsrc + B * P))
    rows.append(("copy mix mode 0 nt (sources cycling)", mix(0, 1, True), nsrc + B * P))
    rows.append(("copy mix mode 1 nt (sources cycling)", mix(1, 1, True), nsrc + B * P))
    for rnd in range(2):
        for name, fn, nbytes in rows:
            if isinstance(fn, tuple):
                setk(fill_mode=fn[1]); f = fill
            else:
                f = fn
            mn, med = loop_us(f)
            print("  %-46s loop avg min %6.2f median %6.2f us -> %5.0f GB/s (%.3f of 8 TB/s)" % (name, mn, med, nbytes / mn / 1e3, nbytes / mn / 1e3 / 8000), flush=True)
    for pad in (0, 8192, 16384, 24576, 40960):
        setk(fill_pad=pad)
        mn, med = loop_us(mix(0))
        print("  copy mix mode 0, unused LDS %5d B:   min %6.2f median %6.2f us" % (pad, mn, med), flush=True)
    setk(fill_pad=0)
