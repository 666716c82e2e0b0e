#!/usr/bin/env python3
"""Consecutive INDEPENDENT batches of a token workload on one stream (in order: what bench.py measures) against two streams taking turns
(the tail of one launch may hide under the head of the next) -- cold regime: > 512 MiB of distinct batches.  Host-clock time per batch
over N launches between two device synchronisations.      two_stream_lab.py [WORKLOAD ...]   (default cfg2 cfg2sf cfg5)"""
import ctypes, importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bsq_bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
import torch
from bioseq_amd import capi
lib = capi.load()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
main = torch.cuda.current_stream()
extra = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
for w in sys.argv[1:] or ["cfg2", "cfg2sf", "cfg5"]:
    b = bench.Batch(w, lib, dev, main)
    n = b.n
    in_bytes = b.total + 8 * (n + 1)
    nb = max(8, -(-(513 << 20) // in_bytes))
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
    torch.cuda.synchronize()
    handles = [ctypes.c_void_p(s.cuda_stream) for s in [main] + extra]

    def run(nstreams, N):
        for i in range(N):
            b.sh = handles[i % nstreams]
            ch, of, out = batches[i % nb]
            b.run(ch, of, out, n)

    for rnd in range(3):
        row = []
        for ns in (1, 2, 4):
            run(ns, 4 * nb); torch.cuda.synchronize()
            N = 3000
            t0 = time.perf_counter(); run(ns, N); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / N
            row.append("%d stream%s %.2f us/batch (frac %.3f)" % (ns, "s" if ns > 1 else "", dt * 1e6, b.algo_bytes / dt / 8e12))
        print("%s round %d  %s" % (w, rnd, " | ".join(row)), flush=True)
    b.sh = handles[0]
    # every batch's result on the last configuration still equals batch 0's rotated (bit-exactness does not depend on the stream)
    run(1, nb); torch.cuda.synchronize()
    axis = 0 if b.batch_first else 1
    for k in range(1, nb):
        assert torch.equal(batches[k][2], torch.roll(batches[0][2], -((k * 4099) % n), dims=axis)), (w, k)
    print(w, "check ok", flush=True)
    del b, batches
    torch.cuda.empty_cache()
