"""The augmentation ALONE on cold batches of cfg5 (8 distinct batches, 584 MB of inputs): one batch per launch (bsq_augment_device), four per
launch (bsq_augment_device_multi), by augment_k and augment_frac; and the token launch alone (four per launch) for scale."""
import os, sys, ctypes, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("b", os.path.join(ROOT, "bench.py")); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
import numpy as np, torch
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
    batches.append((ch, of, torch.empty_like(b.out)))
groups = []
for g in range(nb // 4):
    arr = (capi.Batch * 4)()
    for j in range(4):
        ch, of, out = batches[4 * g + j]
        arr[j].chars, arr[j].offsets, arr[j].B, arr[j].out = ch.data_ptr(), of.data_ptr(), n, out.data_ptr()
    groups.append(arr)
it = [0]
def timed(fn, reps=400):
    return m.timed_loop(fn, reps, 16, stream) * 1e3
for ak in (0, 1):
    capi.check(lib.bsq_tuning_set(b"augment_k", ak))
    for frac in (0.5, 1.0, 0.05):
        def one():
            ch, of, _ = batches[it[0] % nb]
            capi.check(lib.bsq_augment_device(ch.data_ptr(), of.data_ptr(), n, 1, frac, it[0] + 1, b.sh)); it[0] += 1
        def four():
            sd = (ctypes.c_uint64 * 4)(*[it[0] * 4 + j + 1 for j in range(4)])
            capi.check(lib.bsq_augment_device_multi(4, groups[it[0] % 2], 1, frac, sd, b.sh)); it[0] += 1
        print("augment_k %d frac %.2f: one batch per launch %.2f us | four per launch %.2f us per batch" % (ak, frac, timed(one), timed(four, 200) / 4))
capi.check(lib.bsq_tuning_set(b"augment_k", 0))
def tok4():
    capi.check(lib.bsq_tokenize_device_multi(ctypes.byref(b.desc), 4, groups[it[0] % 2], b.P, 1, b.dt_code, b.sh)); it[0] += 1
print("tokens alone, four per launch: %.2f us per batch" % (timed(tok4, 200) / 4))
def both():
    sd = (ctypes.c_uint64 * 4)(*[it[0] * 4 + j + 1 for j in range(4)])
    capi.check(lib.bsq_augment_tokenize_device_multi(ctypes.byref(b.desc), 4, groups[it[0] % 2], b.P, 1, b.dt_code, 1, 0.5, sd, b.sh)); it[0] += 1
print("augment + tokens, four per call: %.2f us per batch" % (timed(both, 200) / 4))
