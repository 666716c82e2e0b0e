"""rocprofv3 target: the augmentation alone on cold cfg5 batches -- 60 launches of bsq_augment_device_multi (4 batches per launch) and 60 of bsq_augment_device."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("b", os.path.join(ROOT, "bench.py")); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
import torch
from bioseq_amd import capi
lib = capi.load()
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
b = m.Batch("cfg5aug", lib, dev, stream)
n, nb = b.n, 8
batches = []
for k in range(nb):
    r = (k * 4099) % n
    c0 = int(b.offsets[r])
    ch = torch.cat([b.d_chars[c0:], b.d_chars[:c0]]) if r else b.d_chars.clone()
    lens = b.d_offs[1:] - b.d_offs[:-1]
    lens = torch.cat([lens[r:], lens[:r]])
    of = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    of[1:] = torch.cumsum(lens, 0)
    batches.append((ch, of))
groups = []
for g in range(nb // 4):
    arr = (capi.Batch * 4)()
    for j in range(4):
        ch, of = batches[4 * g + j]
        arr[j].chars, arr[j].offsets, arr[j].B, arr[j].out = ch.data_ptr(), of.data_ptr(), n, None
    groups.append(arr)
frac = float(os.environ.get("AUG_FRAC", "0.5"))
for it in range(60):
    sd = (ctypes.c_uint64 * 4)(*[it * 4 + j + 1 for j in range(4)])
    capi.check(lib.bsq_augment_device_multi(4, groups[it % 2], 1, frac, sd, b.sh))
for it in range(60):
    ch, of = batches[it % nb]
    capi.check(lib.bsq_augment_device(ch.data_ptr(), of.data_ptr(), n, 1, frac, it + 1, b.sh))
torch.cuda.synchronize()
