#!/usr/bin/env python3
"""int8 one-hots with rows of 3 ... 15 bytes over shapes: the tiled launch, the two-pass form on byte ids and on NIBBLE ids (both expanding
through k_expand_rows1), and the automatic choice -- resident (one batch looped) and COLD (inputs and results cycling over > 600 MB of
distinct buffers).  Every arm is compared with the tiled kernel's result first.      rows1_nib_sweep.py [first_shape [last_shape]]   (DT=f: float32 results; SHIFT=n: results n bytes off a 4-KiB boundary)"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioseq_amd import capi, synth
lib = capi.load()
dev = torch.device("cuda:0")
SHAPES = [("DNA4", (1, 1, 1), 1000000, 150, 150, 160), ("DNA4", (1, 1, 1), 500000, 150, 150, 160), ("DNA4", (1, 1, 1), 250000, 150, 150, 160),
          ("DNA4", (1, 1, 1), 125000, 150, 150, 160), ("DNA4", (1, 1, 1), 2000000, 150, 150, 160), ("DNA4", (0, 0, 0), 1000000, 150, 150, 160),
          ("DNA5", (0, 0, 0), 1000000, 150, 150, 160), ("DNA4", (1, 1, 0), 1000000, 150, 150, 160), ("DNA4", (1, 1, 1), 262144, 30, 512, 512),
          ("DNA4", (1, 1, 1), 65536, 50, 2048, 2048), ("DNA5", (0, 0, 0), 131072, 50, 1024, 1024), ("SEB8", (1, 1, 1), 262144, 30, 512, 512),
          ("SEB14", (0, 0, 0), 131072, 30, 512, 512), ("SEB10", (1, 1, 1), 131072, 30, 512, 512), ("DNA4", (1, 1, 1), 250001, 100, 256, 256),
          ("DNA4", (1, 1, 1), 4000000, 30, 64, 64), ("DNA4", (1, 1, 1), 1500000, 150, 150, 160), ("DNA4", (1, 1, 1), 3000000, 150, 150, 160),
          ("DNA4", (1, 1, 1), 1000000, 200, 250, 256), ("DNA4", (1, 1, 1), 600000, 250, 300, 320),
          ("DNA4", (1, 1, 1), 500000, 300, 380, 384), ("DNA4", (1, 1, 1), 400000, 350, 440, 448), ("DNA4", (1, 1, 1), 500000, 100, 380, 384),
          ("DNA5", (0, 0, 0), 600000, 250, 300, 320), ("DNA4", (0, 0, 0), 800000, 300, 380, 384),
          ("DNA4", (1, 1, 1), 1048576, 150, 150, 160), ("DNA4", (1, 1, 1), 524288, 100, 250, 256), ("DNA4", (1, 1, 1), 262144, 250, 300, 320),
          ("DNA5", (0, 0, 0), 1048576, 150, 150, 160),
          ("DNA4", (1, 1, 1), 299968, 300, 598, 600), ("DNA4", (1, 1, 1), 100032, 500, 1498, 1500), ("DNA4", (1, 1, 1), 262144, 300, 446, 448),
          ("DNA4", (1, 1, 1), 200000, 300, 598, 600), ("DNA4", (1, 1, 1), 131072, 500, 1022, 1024), ("DNA5", (0, 0, 0), 299968, 300, 598, 600),
          # (35 ...) mid-size results at aligned pitches: where should the 192-MB threshold of the one-byte rows be?
          ("DNA4", (1, 1, 1), 16384, 150, 150, 160), ("DNA4", (1, 1, 1), 32768, 150, 150, 160), ("DNA4", (1, 1, 1), 65536, 150, 150, 160),
          ("DNA4", (1, 1, 1), 131072, 150, 150, 160), ("DNA4", (1, 1, 1), 16384, 300, 510, 512), ("DNA4", (1, 1, 1), 32768, 300, 510, 512),
          ("DNA4", (1, 1, 1), 8192, 300, 1022, 1024), ("DNA5", (0, 0, 0), 65536, 150, 150, 160), ("SEB14", (0, 0, 0), 16384, 300, 510, 512),
          ("SEB14", (0, 0, 0), 32768, 300, 510, 512), ("SEB8", (1, 1, 1), 65536, 100, 254, 256)]
DT = os.environ.get("DT", "B")   # destchar: B = int8 (the rows1 forms), f = float32 (k_expand_chunks; the sequence-block cut of very large batches)
SZ = {"B": 1, "f": 4}[DT]
lo_i = int(sys.argv[1]) if len(sys.argv) > 1 else 0
hi_i = int(sys.argv[2]) if len(sys.argv) > 2 else len(SHAPES)
ARMS = [("tile", dict(onehot_path=1)), ("2p-bytes", dict(onehot_path=2, raw_nibbles=1)), ("2p-nib", dict(onehot_path=2, raw_nibbles=2)), ("auto", dict())]


def setk(**kw):
    for k in ("onehot_path", "raw_nibbles"):
        capi.check(lib.bsq_tuning_set(k.encode(), int(kw.get(k, 0))))


for si, (key, flags, B, lo, hi, P) in list(enumerate(SHAPES))[lo_i:hi_i]:
    letters = synth.AA if key[0] != "D" or key == "DAYHOFF" else "ACGT"
    chars, offs = synth.synth_packed(2000 + si, B, lo, hi, letters)
    desc = capi.make_desc(key, *flags)
    C = lib.bsq_alphabet_size(ctypes.byref(desc))
    dt = ctypes.c_int(0); capi.check(lib.bsq_dtype_from_destchar(DT.encode(), ctypes.byref(dt)))
    ob = P * B * C * SZ
    algo = int(offs[-1]) + 8 * (B + 1) + ob
    K = max(2, -(-600_000_000 // algo) + 1)
    ins = [(torch.from_numpy(chars).to(dev), torch.from_numpy(offs).to(dev)) for _ in range(K)]
    SHIFT = int(os.environ.get("SHIFT", "0"))   # results that many bytes off a 4-KiB boundary (torch aligns sub-allocations to 512 bytes only)
    bufs = [torch.empty(ob + 8192, dtype=torch.uint8, device=dev) for _ in range(K)]
    outs = [b[(-b.data_ptr()) % 4096 + SHIFT:][:ob] for b in bufs]
    ref = torch.empty(ob, dtype=torch.uint8, device=dev)

    def run(i):
        dch, dof = ins[i % K]
        capi.check(lib.bsq_onehot_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), None, B, P, dt, outs[i % K].data_ptr(), None))

    res = []
    for name, kw in ARMS:
        setk(**kw)
        outs[0].fill_(5); run(0); torch.cuda.synchronize()
        if name == "tile": ref.copy_(outs[0])
        else: assert torch.equal(outs[0], ref), (key, name)
    for name, kw in ARMS:
        setk(**kw)
        t_res, t_cold = [], []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            run(0); a.record()
            for _ in range(6): run(0)
            b.record(); torch.cuda.synchronize(); t_res.append(a.elapsed_time(b) / 6)
            n = 3 * K
            for i in range(K): run(i)
            a.record()
            for i in range(n): run(i)
            b.record(); torch.cuda.synchronize(); t_cold.append(a.elapsed_time(b) / n)
        r, c = np.median(t_res), np.median(t_cold)
        res.append("%s %.1f us (%.3f) cold %.1f us (%.3f)" % (name, r * 1e3, algo / r / 8e9, c * 1e3, algo / c / 8e9))
    kn = lib.bsq_onehot_kernel_name(ctypes.byref(desc), B, P, dt)
    print("%-6s %s B=%7d P=%4d C=%2d out=%5.2f GB K=%d | %s | auto = %s" % (key, flags, B, P, C, ob / 1e9, K, " | ".join(res), kn.decode() if kn else ""), flush=True)
    del ins, outs, bufs, ref
    torch.cuda.empty_cache()
setk()
