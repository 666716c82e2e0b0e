"""rocprofv3 target: the raw-id pass of cfg4 (nibble ids) alone, 40 launches on ONE resident batch, then 40 cycling over 6 distinct batches."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from bioseq_amd import capi, synth
lib = capi.load()
dev = torch.device("cuda:0")
c = synth.CONFIGS["cfg4"]
chars, offs = synth.synth_packed(c["seed"], c["n"], c["lo"], c["hi"], c["letters"])
desc = capi.make_desc(c["key"], c["eos"], c["bos"], c["padchar"])
B, P = c["n"], c["padlen"]
d_offs = torch.from_numpy(offs).to(dev)
copies = [torch.from_numpy(chars).to(dev) for _ in range(6)]
# the library's own two-pass call with a tiny expansion is not separable: use the one-hot entry and read the raw kernel's rows from the trace
out = torch.empty((P, B, 7), dtype=torch.int8, device=dev)
for it in range(40):
    capi.check(lib.bsq_onehot_device(ctypes.byref(desc), copies[0].data_ptr(), d_offs.data_ptr(), None, B, P, capi.I8, out.data_ptr(), None))
torch.cuda.synchronize()
for it in range(60):
    capi.check(lib.bsq_onehot_device(ctypes.byref(desc), copies[it % 6].data_ptr(), d_offs.data_ptr(), None, B, P, capi.I8, out.data_ptr(), None))
torch.cuda.synchronize()
