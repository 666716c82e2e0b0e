"""Target of tile_placement_pmc.sh: the forced tiled one-hot on dna4_i16_512, 10 launches into each of several result ALLOCATIONS of one process
(a first big one, then fresh ones behind fillers -- tile_placement.py: 295-340 us in the first kind, 380-388 in the second), then the two-pass form
into the last one.  Prints the events time per allocation; the profiler passes attribute launches to allocations by their ordinal."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bioseq_amd import capi, synth
lib = capi.load()
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
sh = ctypes.c_void_p(stream.cuda_stream)
key, (bos, eos, pad), B, P, dch, lo, hi = ("DNA4", (1, 1, 1), 262144, 512, "h", 50, 510)
desc = capi.make_desc(key, eos, bos, pad)
C = lib.bsq_alphabet_size(ctypes.byref(desc))
dt = ctypes.c_int(0)
capi.check(lib.bsq_dtype_from_destchar(dch.encode(), ctypes.byref(dt)))
chars, offs = synth.synth_packed(1234, B, lo, hi, "ACGT")
d_offs = torch.from_numpy(offs).to(dev)
copies = [torch.from_numpy(chars).to(dev) for _ in range(6)]
total = P * B * C * 2
N = 10
def run(ptr, knob):
    capi.check(lib.bsq_tuning_set(b"onehot_path", knob))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(N):
        if i == 2:
            a.record(stream)
        capi.check(lib.bsq_onehot_device(ctypes.byref(desc), copies[i % 6].data_ptr(), d_offs.data_ptr(), None, B, P, dt, ctypes.c_void_p(ptr), sh))
    b.record(stream)
    torch.cuda.synchronize()
    capi.check(lib.bsq_tuning_set(b"onehot_path", 0))
    return a.elapsed_time(b) / (N - 2) * 1e3
for label, filler_mb in (("first", 0), ("filler3", 3), ("filler64", 64), ("filler513", 513), ("again0", 0)):
    filler = torch.empty(filler_mb << 20, dtype=torch.uint8, device=dev) if filler_mb else None
    out_t = torch.empty(total + ((96 << 20) if label == "first" else 0), dtype=torch.uint8, device=dev)
    print("allocation %-9s at 0x%x: tile %.1f us" % (label, out_t.data_ptr(), run(out_t.data_ptr(), 1)), flush=True)
    if label == "again0":
        print("allocation %-9s two-pass %.1f us" % (label, run(out_t.data_ptr(), 2)), flush=True)
    del out_t, filler
    torch.cuda.empty_cache()
