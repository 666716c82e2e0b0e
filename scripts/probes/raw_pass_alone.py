"""The raw-id pass alone (bsq_raw_tokens_device: byte ids) on cfg4's batch and on cfg3's: looped over ONE resident batch and cycling over distinct
batches, with the paired tail on (knob 0) and off (knob 1), interleaved; HIP events."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bioseq_amd import capi, synth
lib = capi.load()
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
sh = ctypes.c_void_p(stream.cuda_stream)
def timed(fn, n, warm):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(n): fn()
    b.record(stream); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for cfgname, ncopies in (("cfg4", 6),):
    c = synth.CONFIGS[cfgname]
    chars, offs = synth.synth_packed(c["seed"], c["n"], c["lo"], c["hi"], c["letters"])
    desc = capi.make_desc(c["key"], c["eos"], c["bos"], c["padchar"])
    B, P = c["n"], c["padlen"]
    pitch = (B + 255) // 256 * 256
    d_offs = torch.from_numpy(offs).to(dev)
    copies = [torch.from_numpy(chars).to(dev) for _ in range(ncopies)]
    toks = [torch.empty(P * pitch, dtype=torch.uint8, device=dev) for _ in range(ncopies)]
    it = [0]
    def resident():
        capi.check(lib.bsq_raw_tokens_device(ctypes.byref(desc), copies[0].data_ptr(), d_offs.data_ptr(), None, B, P, toks[0].data_ptr(), pitch, sh))
    def cycling():
        k = it[0] % ncopies; it[0] += 1
        capi.check(lib.bsq_raw_tokens_device(ctypes.byref(desc), copies[k].data_ptr(), d_offs.data_ptr(), None, B, P, toks[k].data_ptr(), pitch, sh))
    for rnd in range(3):
        for knob in (0, 1):
            capi.check(lib.bsq_tuning_set(b"tokens_pb8_pair", knob))
            print("%s round %d pair knob %d: raw pass alone resident %.2f us | cycling over %d batches %.2f us" % (cfgname, rnd, knob, timed(resident, 100, 20), ncopies, timed(cycling, 120, 24)))
    capi.check(lib.bsq_tuning_set(b"tokens_pb8_pair", 0))
