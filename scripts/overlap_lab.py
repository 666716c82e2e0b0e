#!/usr/bin/env python3
"""Can the raw-id pass of the two-pass one-hot hide under the expansion?  (round 5)
The one-launch form lost in round 4 (profiles/r04/onehot_fused_one_launch_lost.txt: bulk producer -> consumer traffic between XCDs inside
ONE launch does not pay).  Here the passes stay separate kernels -- kernel boundaries do the release / acquire -- but run on TWO streams:
    S:  raw(rows 0 .. P1) -> expand(rows 0 .. P1) -> [wait R] -> expand(rows P1 .. P)
    B:  [wait fork] -> raw(rows P1 .. P) -> record R
so only the first slice's raw pass is exposed.  This lab needs no kernel change: it TIMES that schedule with the public entries
(bsq_raw_tokens_device / bsq_onehot_from_raw_tokens_device), the expansions reading a correct id matrix made beforehand and the raw
passes writing dummies of the right size (first slice: a padlen-P1 call; second stream: the whole matrix -- a little MORE work than the
real schedule's remainder).  Arms: serial (the product today), expansion alone, raw alone, overlap with P1 in a list, and the
expansion in two launches without any raw pass (the price of the extra kernel boundary).
    overlap_lab.py [workloads]     workloads: comma list of cfg3,cfg3b,cfg4f,cfg4b (default: all)"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioseq_amd import capi, synth
lib = capi.load()
dev = torch.device("cuda:0")
WL = {"cfg3": ("AMINO20", (1, 1, 1), 65536, 50, 1022, 1024, b"f", synth.AA, (64, 128, 256)),
      "cfg3b": ("AMINO20", (1, 1, 1), 65536, 50, 1022, 1024, b"B", synth.AA, (64, 128, 256)),
      "cfg4f": ("DNA4", (1, 1, 1), 1000000, 150, 150, 160, b"f", "ACGT", (16, 32, 64)),
      "cfg4b": ("DNA4", (1, 1, 1), 1000000, 150, 150, 160, b"B", "ACGT", (16, 32, 64))}
names = sys.argv[1].split(",") if len(sys.argv) > 1 else list(WL)
if os.environ.get("ONEHOT_PATH"): capi.check(lib.bsq_tuning_set(b"onehot_path", int(os.environ["ONEHOT_PATH"])))
for name in names:
    key, flags, B, lo, hi, P, dc, letters, P1s = WL[name]
    chars, offs = synth.synth_packed(77, B, lo, hi, letters)
    desc = capi.make_desc(key, *flags)
    C = lib.bsq_alphabet_size(ctypes.byref(desc))
    dt = ctypes.c_int(0); capi.check(lib.bsq_dtype_from_destchar(dc, ctypes.byref(dt)))
    sz = lib.bsq_dtype_size(dt)
    dch, dof = torch.from_numpy(chars).to(dev), torch.from_numpy(offs).to(dev)
    pitch = (B + 255) // 256 * 256
    ids = torch.empty(P * pitch, dtype=torch.uint8, device=dev)      # the correct id matrix
    dummyA = torch.empty(P * pitch, dtype=torch.uint8, device=dev)   # what the timed raw passes write
    dummyB = torch.empty(P * pitch, dtype=torch.uint8, device=dev)
    rowb = B * C * sz
    out = torch.empty(P * rowb, dtype=torch.uint8, device=dev)
    ref = torch.empty(P * rowb, dtype=torch.uint8, device=dev)
    algo = int(offs[-1]) + 8 * (B + 1) + P * rowb
    S, Bs = torch.cuda.Stream(), torch.cuda.Stream()
    sp = lambda st: ctypes.c_void_p(st.cuda_stream)
    def raw(dst, p, st): capi.check(lib.bsq_raw_tokens_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), None, B, p, dst.data_ptr(), pitch, sp(st)))
    def expand(p0, p1, st): capi.check(lib.bsq_onehot_from_raw_tokens_device(ids.data_ptr() + p0 * pitch, pitch, B, p1 - p0, C, dt, out.data_ptr() + p0 * rowb, sp(st)))
    def whole(st): capi.check(lib.bsq_onehot_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), None, B, P, dt, out.data_ptr(), sp(st)))
    with torch.cuda.stream(S):
        whole(S); S.synchronize(); ref.copy_(out); raw(ids, P, S); out.fill_(7); expand(0, P // 2, S); expand(P // 2, P, S); S.synchronize()
        assert torch.equal(out, ref), "expansion in two launches differs"
    def serial(): raw(dummyA, P, S); expand(0, P, S)
    def overlap(P1):
        def f():
            fork = torch.cuda.Event(); fork.record(S); Bs.wait_event(fork)
            raw(dummyB, P, Bs); r = torch.cuda.Event(); r.record(Bs)
            raw(dummyA, P1, S); expand(0, P1, S); S.wait_event(r); expand(P1, P, S)
        return f
    def split_only(P1): return lambda: (expand(0, P1, S), expand(P1, P, S))
    # the expansion's occupancy cap is unused dynamic LDS filling the CU (3 x 52 KiB / 5 x 32 KiB): a raw-pass workgroup (17.5 KiB) cannot
    # become resident beside it.  `roomy` = the same number of expansion workgroups per CU with 20-22 KiB left over for one.
    roomy = 30720 if C * sz >= 64 else 12288
    def with_pad(pad, f):
        def g():
            capi.check(lib.bsq_tuning_set(b"expand_pad", pad)); f(); capi.check(lib.bsq_tuning_set(b"expand_pad", 0))
        return g
    arms = [("product call", lambda: whole(S)), ("serial raw + expand", serial), ("expand alone", lambda: expand(0, P, S)), ("raw alone", lambda: raw(dummyA, P, S)),
            ("serial, roomy pad", with_pad(roomy, serial))]
    for P1 in P1s:
        arms += [("overlap P1=%d" % P1, overlap(P1)), ("overlap P1=%d, roomy pad" % P1, with_pad(roomy, overlap(P1))), ("expand in two launches, P1=%d" % P1, split_only(P1))]
    res = {n: [] for n, _ in arms}
    for rnd in range(4):
        for n, f in arms:
            with torch.cuda.stream(S):
                for _ in range(3): f()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(S)
                for _ in range(10): f()
                b.record(S); torch.cuda.synchronize()
            if rnd: res[n].append(a.elapsed_time(b) / 10)
    print("%s B=%d P=%d C=%d %s out=%.2f GB" % (name, B, P, C, dc.decode(), P * rowb / 1e9))
    for n, _ in arms:
        t = float(np.median(res[n])) * 1e3
        print("   %-36s %8.1f us   frac of 8 TB/s %.3f" % (n, t, algo / t / 8e6), flush=True)
    del ids, dummyA, dummyB, out, ref, dch, dof
