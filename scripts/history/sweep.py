#!/usr/bin/env python3
"""A/B sweep of kernel tuning knobs on one workload, interleaved rounds in ONE process.
   python scripts/sweep.py [workload] -- prints median/min ms per variant."""
import ctypes, itertools, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bioseq_amd
from bioseq_amd import capi, synth

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
WL = {"cfg3": ("cfg3", "onehot", "f"), "cfg4f": ("cfg4", "onehot", "f"), "cfg4b": ("cfg4", "onehot", "B"),
      "cfg2": ("cfg2", "tokenize", "B"), "cfg2sf": ("cfg2", "tokenize_sf", "B"), "cfg5": ("cfg5", "tokenize", "B")}
cfgname, op, destchar = WL[wl]
cfg = synth.CONFIGS[cfgname]
dev = torch.device("cuda:0")
lib = capi.load()
chars, offs = synth.synth_packed(cfg["seed"], cfg["n"], cfg["lo"], cfg["hi"], cfg["letters"])
n, P = cfg["n"], cfg["padlen"]
desc = capi.make_desc(cfg["key"], cfg["eos"], cfg["bos"], cfg["padchar"])
C = lib.bsq_alphabet_size(ctypes.byref(desc))
dt = ctypes.c_int(0); capi.check(lib.bsq_dtype_from_destchar(destchar.encode(), ctypes.byref(dt)))
sz = lib.bsq_dtype_size(dt)
dch, dof = torch.from_numpy(chars).to(dev), torch.from_numpy(offs).to(dev)
out_bytes = P * n * (C if op == "onehot" else 1) * sz
out = torch.empty(out_bytes, dtype=torch.uint8, device=dev)
ref = torch.empty(out_bytes, dtype=torch.uint8, device=dev)
algo = int(offs[-1]) + 8 * (n + 1) + out_bytes

def run():
    if op == "onehot":
        capi.check(lib.bsq_onehot_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), None, n, P, dt, out.data_ptr(), None))
    else:
        capi.check(lib.bsq_tokenize_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), n, P, int(op == "tokenize"), dt, out.data_ptr(), None))

def setk(**kw):
    for k in ("nt_stores", "onehot_tb", "tile_order", "onehot_path", "chunks_cpw"):
        capi.check(lib.bsq_tuning_set(k.encode(), int(kw.get(k, 0))))

variants = [dict()]
if op == "onehot":
    variants = [dict(onehot_path=1, nt_stores=nt) for nt in (0, 1)] + \
               [dict(onehot_path=2, nt_stores=nt) for nt in (0, 1)] + \
               [dict(onehot_path=3, chunks_cpw=c, nt_stores=nt) for c in (1, 2, 4) for nt in (0, 1)]
setk(); run(); torch.cuda.synchronize(); ref.copy_(out)
times = {i: [] for i in range(len(variants))}
for rnd in range(6):
    for i, v in enumerate(variants):
        setk(**v)
        out.fill_(3)
        run(); torch.cuda.synchronize()
        if rnd == 0:
            assert torch.equal(out, ref), ("variant changes results", v)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): run()
        b.record(); torch.cuda.synchronize()
        times[i].append(a.elapsed_time(b) / 5)
print("workload", wl, "algorithmic bytes", algo)
for i, v in enumerate(variants):
    t = np.array(times[i]); med = float(np.median(t))
    print("%-60s median %.4f ms  min %.4f ms  -> %.0f GB/s (%.1f%% of 8 TB/s)" % (json.dumps(v), med, t.min(), algo / med / 1e6, algo / med / 1e6 / 80))
# fill yardsticks
setk()
fb = (out_bytes // 16) * 16
for mode in range(5):
    capi.check(lib.bsq_tuning_set(b"fill_mode", mode))
    ts = []
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        capi.check(lib.bsq_fill_device(out.data_ptr(), fb, 0, None))
        a.record()
        for _ in range(5): capi.check(lib.bsq_fill_device(out.data_ptr(), fb, 0, None))
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 5)
    print("fill_mode %d: median %.4f ms -> %.0f GB/s" % (mode, np.median(ts), fb / np.median(ts) / 1e6))
ts = []
o32 = out[:fb].view(torch.float32)
for _ in range(4):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    o32.fill_(1.0); a.record()
    for _ in range(5): o32.fill_(1.0)
    b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 5)
print("torch fill_: median %.4f ms -> %.0f GB/s" % (np.median(ts), fb / np.median(ts) / 1e6))
capi.check(lib.bsq_tuning_set(b"fill_mode", 0))
