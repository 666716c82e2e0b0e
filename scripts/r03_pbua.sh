#!/bin/bash
# k_tokens_pb8_fast on rows that are only element-aligned: parity, then the odd-shape sweep with (default) and without (knob 3) its UA form
OUT=gpurun_out/r03pbua; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_tokens_seqfirst.py -m gpu -x -q 2>&1 | tail -3 | tee $OUT/tests.txt
python3 - <<'PY' 2>/dev/null | tee $OUT/odd_shapes.txt
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from bioseq_amd import capi, synth
lib = capi.load(); dev = torch.device("cuda:0")
desc = capi.make_desc("AMINO20", 0, 0, 0)
for B, P in ((65536, 1024), (65000, 1024), (65000, 1001), (65537, 1024), (100001, 512), (250001, 256)):
    chars, offs = synth.synth_packed(7, B, 50 if P > 300 else 10, P - 2, synth.AA)
    dch, dof = torch.from_numpy(chars).to(dev), torch.from_numpy(offs).to(dev)
    for dc in "bh":
        dt = ctypes.c_int(0); capi.check(lib.bsq_dtype_from_destchar(dc.encode(), ctypes.byref(dt))); sz = lib.bsq_dtype_size(dt)
        out = torch.empty(B * P * sz + 64, dtype=torch.uint8, device=dev)
        res = []
        for knob in (3, 4):
            capi.check(lib.bsq_tuning_set(b"tokens_pb8", knob))
            def run(): capi.check(lib.bsq_tokenize_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), B, P, 0, dt, out.data_ptr(), None))
            for _ in range(20): run()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(200): run()
            b.record(); torch.cuda.synchronize()
            res.append(a.elapsed_time(b) / 200 * 1e3)
        capi.check(lib.bsq_tuning_set(b"tokens_pb8", 0))
        print("B=%6d P=%4d destchar %s (P,B): aligned-only %6.1f us | with the UA form %6.1f us" % (B, P, dc, res[0], res[1]), flush=True)
PY
