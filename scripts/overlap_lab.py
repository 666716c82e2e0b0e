#!/usr/bin/env python3
"""Would overlapping the token pass (vector-ALU / LDS bound) of one slice with the expansion (HBM-write bound) of another
pay?  Through the C ABI's two passes on two streams: N steps sequential on one stream vs the token pass of step i + 1 on
a second stream while step i expands.  cfg3 and cfg4 f32."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioseq_amd import capi, synth
lib = capi.load()
dev = torch.device("cuda:0")
for name in ("cfg3", "cfg4"):
    c = synth.CONFIGS[name]
    chars, offs = synth.synth_packed(c["seed"], c["n"], c["lo"], c["hi"], c["letters"])
    dch, dof = torch.from_numpy(chars).to(dev), torch.from_numpy(offs).to(dev)
    desc = capi.make_desc(c["key"], c["eos"], c["bos"], c["padchar"])
    B, P = c["n"], c["padlen"]
    C = lib.bsq_alphabet_size(ctypes.byref(desc))
    pitch = (B + 255) // 256 * 256
    tok = [torch.empty(P * pitch, dtype=torch.uint8, device=dev) for _ in range(2)]
    out = torch.empty(P * B * C * 4, dtype=torch.uint8, device=dev)
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    def tokens(i, s): capi.check(lib.bsq_raw_tokens_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), None, B, P, tok[i].data_ptr(), pitch, s.cuda_stream))
    def expand(i, s): capi.check(lib.bsq_onehot_from_raw_tokens_device(tok[i].data_ptr(), pitch, B, P, C, capi.F32, out.data_ptr(), s.cuda_stream))
    N = 20
    def sequential():
        for i in range(N):
            tokens(i & 1, sA); expand(i & 1, sA)
    def overlapped():
        ev_tok = [torch.cuda.Event() for _ in range(N + 1)]
        ev_exp = [torch.cuda.Event() for _ in range(N + 1)]
        tokens(0, sB); ev_tok[0].record(sB)
        for i in range(N):
            sA.wait_event(ev_tok[i])
            expand(i & 1, sA); ev_exp[i].record(sA)
            if i + 1 < N:
                if i >= 1: sB.wait_event(ev_exp[i - 1])  # the scratch being rewritten was last read by expansion i - 1
                tokens((i + 1) & 1, sB); ev_tok[i + 1].record(sB)
    for label, fn in (("sequential", sequential), ("overlapped", overlapped), ("sequential", sequential), ("overlapped", overlapped)):
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(sA); sB.wait_event(a)
        fn()
        sA.wait_stream(sB); b.record(sA); torch.cuda.synchronize()
        print("%s %s: %.4f ms per step" % (name, label, a.elapsed_time(b) / N), flush=True)
