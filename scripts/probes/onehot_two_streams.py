"""Would the raw-id pass of the two-pass one-hot hide under the expansion?  Upper bound without touching the library:
whole one-hot calls (raw pass + expansion) of INDEPENDENT batches alternating between two streams -- the raw pass of batch k + 1 then runs beside
the expansion of batch k -- against the same calls in order on one stream.  Cold batches (cfg4b / cfg4f: distinct inputs > 512 MiB; outputs in two
buffers).  Host clock between two synchronisations, per batch."""
import os, sys, ctypes, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("b", os.path.join(ROOT, "bench.py")); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
import numpy as np, torch
from bioseq_amd import capi
lib = capi.load()
dev = torch.device("cuda:0")
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
for w in sys.argv[1:] or ["cfg4b", "cfg3b", "cfg4f"]:
    b = m.Batch(w, lib, dev, main)
    n = b.n
    nb = 4
    ins = []
    for k in range(nb):
        r = (k * 4099) % n
        c0 = int(b.offsets[r])
        ch = torch.cat([b.d_chars[c0:], b.d_chars[:c0]]) if r else b.d_chars.clone()
        lens = b.d_offs[1:] - b.d_offs[:-1]
        lens = torch.cat([lens[r:], lens[:r]])
        of = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        of[1:] = torch.cumsum(lens, 0)
        ins.append((ch, of))
    outs = [b.out, torch.empty_like(b.out)]
    handles = [ctypes.c_void_p(main.cuda_stream), ctypes.c_void_p(side.cuda_stream)]

    def run(nl, two):
        for i in range(nl):
            ch, of = ins[i % nb]
            b.sh = handles[i & 1] if two else handles[0]
            b.run(ch, of, outs[i & 1], n)

    res = {}
    for two in (False, True, False, True):
        run(8, two)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(60, two)
        torch.cuda.synchronize()
        res.setdefault(two, []).append((time.perf_counter() - t0) / 60 * 1e3)
    b.sh = handles[0]
    one, two = min(res[False]), min(res[True])
    print("%-6s %s: in order %.4f ms (frac %.3f) | two streams %.4f ms (frac %.3f) | %+.1f %%" % (
        w, b.kernel_name(), one, b.algo_bytes / one / 1e6 / 8000, two, b.algo_bytes / two / 1e6 / 8000, (one / two - 1) * 100))
    del b, ins, outs
    torch.cuda.empty_cache()
