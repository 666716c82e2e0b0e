"""Target of tile_clock.sh: two shapes where the tiled one-hot kernel's time moved by 25 % between boxes (profiles/r06/dispatch_check.txt),
each forced through `k_onehot_tile` (knob onehot_path = 1) and through the two-pass form (= 2) -- 30 launches of each, cycling over distinct
copies of the input -- plus one ALU-only torch kernel of fixed work (a clock yardstick that touches no memory to speak of)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bioseq_amd import capi, synth
lib = capi.load()
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
sh = ctypes.c_void_p(stream.cuda_stream)
# (scripts/dispatch_shapes.json: dna4_i16_512 -- 311 us on one box, 387-390 on another -- and dna5_long1024 -- 136 us in round 5, 170-183 in round 6)
SHAPES = [("dna4_i16_512", "DNA4", (1, 1, 1), 262144, 512, "h", 50, 510), ("dna5_long1024", "DNA5", (0, 0, 0), 131072, 1024, "b", 200, 1024)]
for name, key, (bos, eos, pad), B, P, dch, lo, hi in SHAPES:
    desc = capi.make_desc(key, eos, bos, pad)
    C = lib.bsq_alphabet_size(ctypes.byref(desc))
    dt = ctypes.c_int(0)
    capi.check(lib.bsq_dtype_from_destchar(dch.encode(), ctypes.byref(dt)))
    tdt = {0: torch.int8, 1: torch.int16}[dt.value]
    chars, offs = synth.synth_packed(1234, B, lo, hi, "ACGT")
    d_offs = torch.from_numpy(offs).to(dev)
    copies = [torch.from_numpy(chars).to(dev) for _ in range(4)]
    out_t = torch.empty((P, B, C), dtype=tdt, device=dev)
    ev = {}
    for knob, label in ((1, "tile"), (2, "two_pass"), (1, "tile"), (2, "two_pass")):
        capi.check(lib.bsq_tuning_set(b"onehot_path", knob))
        for i in range(6):
            capi.check(lib.bsq_onehot_device(ctypes.byref(desc), copies[i % 4].data_ptr(), d_offs.data_ptr(), None, B, P, dt, out_t.data_ptr(), sh))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for i in range(24):
            capi.check(lib.bsq_onehot_device(ctypes.byref(desc), copies[i % 4].data_ptr(), d_offs.data_ptr(), None, B, P, dt, out_t.data_ptr(), sh))
        b.record(stream)
        torch.cuda.synchronize()
        ev.setdefault(label, []).append(a.elapsed_time(b) / 24 * 1e3)
    capi.check(lib.bsq_tuning_set(b"onehot_path", 0))
    print("%s: events, us per call: %s" % (name, "  ".join("%s %s" % (k, "/".join("%.1f" % x for x in v)) for k, v in ev.items())), flush=True)
    del copies, out_t
    torch.cuda.empty_cache()
# the yardstick: 64 dependent transcendental passes over 4 M floats that stay in L2 / Infinity Cache: issue-bound, scales with the shader clock
x = torch.rand(1 << 22, device=dev)
for _ in range(3):
    y = x
    for _ in range(8):
        y = torch.erfinv(y * 0.5)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(stream)
for _ in range(20):
    y = x
    for _ in range(8):
        y = torch.erfinv(y * 0.5)
b.record(stream)
torch.cuda.synchronize()
print("yardstick: 8 x (mul, erfinv) over 4 M floats: %.1f us" % (a.elapsed_time(b) / 20 * 1e3))
