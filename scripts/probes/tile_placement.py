"""`k_onehot_tile`'s 25 % "box-to-box" variance (profiles/r06/dispatch_check.txt): the same box gives 305 us in one PROCESS and 380 us in the next
(profiles/r06/tile_kernel_variance.txt).  What moves it inside one process?  The result tensor at different virtual offsets of one big allocation,
fresh allocations, the input resident or cold -- forced tiled kernel and forced two-pass beside it, HIP events."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bioseq_amd import capi, synth
lib = capi.load()
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
sh = ctypes.c_void_p(stream.cuda_stream)
SHAPES = {"dna4_i16_512": ("DNA4", (1, 1, 1), 262144, 512, "h", 50, 510),      # automatic choice: two-pass
          "dna4c4_i16": ("DNA4", (0, 0, 0), 1000000, 160, "h", 150, 150),        # automatic choice: the tiled kernel (8-byte rows of int16, 1.28 GB)
          "dna5_i16": ("DNA5", (0, 0, 0), 1000000, 160, "h", 150, 150)}          # automatic choice: the tiled kernel (10-byte rows, 1.6 GB)
name = sys.argv[1] if len(sys.argv) > 1 else "dna4_i16_512"
key, (bos, eos, pad), B, P, dch, lo, hi = SHAPES[name]
desc = capi.make_desc(key, eos, bos, pad)
C = lib.bsq_alphabet_size(ctypes.byref(desc))
dt = ctypes.c_int(0)
capi.check(lib.bsq_dtype_from_destchar(dch.encode(), ctypes.byref(dt)))
chars, offs = synth.synth_packed(1234, B, lo, hi, "ACGT")
d_offs = torch.from_numpy(offs).to(dev)
copies = [torch.from_numpy(chars).to(dev) for _ in range(6)]
total = P * B * C * 2
print("%s: result %.2f GB, row pitch %d bytes (= %.3f MiB), input %.1f MB per copy" % (name, total / 1e9, B * C * 2, B * C * 2 / 2**20, chars.size / 1e6))

def timed(out_ptr, knob, ncopies, n=12, warm=4):
    capi.check(lib.bsq_tuning_set(b"onehot_path", knob))
    def step(i):
        capi.check(lib.bsq_onehot_device(ctypes.byref(desc), copies[i % ncopies].data_ptr(), d_offs.data_ptr(), None, B, P, dt, ctypes.c_void_p(out_ptr), sh))
    for i in range(warm):
        step(i)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for i in range(n):
        step(i)
    b.record(stream)
    torch.cuda.synchronize()
    capi.check(lib.bsq_tuning_set(b"onehot_path", 0))
    return a.elapsed_time(b) / n * 1e3

big = torch.empty(total + (96 << 20), dtype=torch.uint8, device=dev)
base = big.data_ptr()
print("one allocation at 0x%x; the result at offsets of it (tile cold / tile input-resident / two-pass cold, us):" % base)
for off in (0, 4096, 65536, 1 << 20, (2 << 20) + 4096, 3 << 20, (16 << 20) + 8192, 33 << 20, 64 << 20, 0):
    print("  +%-10d  tile %6.1f  tile(resident input) %6.1f  two-pass %6.1f" % (off, timed(base + off, 1, 6), timed(base + off, 1, 1), timed(base + off, 2, 6)), flush=True)
del big
torch.cuda.empty_cache()
print("fresh allocations (each after empty_cache; a filler of a different size in front):")
for filler_mb in (0, 1, 3, 64, 513, 0):
    filler = torch.empty(max(filler_mb, 0) << 20, dtype=torch.uint8, device=dev) if filler_mb else None
    out_t = torch.empty(total, dtype=torch.uint8, device=dev)
    print("  filler %4d MB, result at 0x%x:  tile %6.1f  two-pass %6.1f" % (filler_mb, out_t.data_ptr(), timed(out_t.data_ptr(), 1, 6), timed(out_t.data_ptr(), 2, 6)), flush=True)
    del out_t, filler
    torch.cuda.empty_cache()
