#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2|cfg4f|cfg4b|cfg5]
                    [--scaling weak|strong] [--gather K]

A "step" is ONE pass of the hot path over one batch of synthetic input that is already resident
in HBM (packed chars + offsets on the device, output preallocated): for the default workload
(BASELINE.json configs[2], "cfg3") that is `batch_onehot_encode` of 65 536 AMINO20 sequences,
len ~ U(50,1024), padlen 1024 -> float32 (1024, 65536, 20) = 5.37 GB, through the C ABI entry
point bsq_onehot_device.  With N > 1 (launched by torch.distributed.run, one rank per GPU) every
rank encodes its own 65 536-sequence shard of an N x 65 536 batch -- sequences are independent, so
there is no data-path collective (weak scaling) -- and `value` is the whole-job rate.  `--scaling strong`
splits ONE batch of the workload's size over the ranks instead (sharding.shard_bounds: BASELINE config 4 is
"1M reads, 1 vs 8 GPU shard + RCCL gather" -> --workload cfg4f --scaling strong --gather 3); `--gather K`
additionally times the whole-batch assembly over xGMI in each of its forms, never as part of `value`.

Rank 0 prints ONE JSON line.  Extra objects:
  roofline      algorithmic bytes per launch / average kernel duration (HIP events on the launch
                stream: one pair of events around the K timed launches; a second, untimed pass
                with one pair per step gives min / median) against the 8 TB/s HBM3E peak; `traffic` = measured HBM bytes per launch
                from the committed rocprofv3 PMC pass (profiles/traffic.json) or null.
  cpu_baseline  the reference's CPU path on this box's host cores, same batch (N=1, rank 0 only):
                oracle/_ref (the reference's own C++ compiled in place, kind "reference") when that
                prebuilt module is present, else the C port in oracle/ (kind "port").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

WORKLOADS = {
    #        synth config, op, destchar, batch_first
    "cfg3": ("cfg3", "onehot", "f", False),
    "cfg3bcl": ("cfg3", "onehot_bcl", "f", False),  # channels-first (B,C,P) written directly (loader layout)
    "cfg1oh": ("cfg1", "onehot", "f", False),     # BASELINE configs[0]'s batch as a one-hot: tiny, for smoke runs of the N > 1 path
    "cfg2": ("cfg2", "tokenize", "B", True),
    "cfg2sf": ("cfg2", "tokenize", "B", False),   # the reference's DEFAULT layout of batch_tokenize: (padlen, batch)
    "cfg4f": ("cfg4", "onehot", "f", False),
    "cfg4b": ("cfg4", "onehot", "B", False),
    "cfg5": ("cfg5", "tokenize", "B", True),
    "cfg5aug": ("cfg5", "augment+tokenize", "B", True),  # BASELINE config 5: BLOSUM62 augmentation, then SEB8 tokens
}


def baseline_metric():
    """The headline metric string of BASELINE.json (the default workload measures exactly that config)."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except Exception:
        return "Gseq-chars/s + GB/s one-hot written, 64k\u00d71024 AMINO20, 1/2/4/8 GPU"


def cpu_baseline(cfg, op, destchar, batch_first, chars, offsets, cap_threads=None):
    """Time the reference CPU path on this box's host cores (bounded: two calls): all cores on the full batch --
    the headline `value` -- and nthreads=1 (BASELINE.md section 4 asks for both) on the first eighth of it."""
    from bioseq_amd import synth
    from oracle import oracle as O  # checker / baseline only -- never on the product path
    cores = os.cpu_count() or 1
    nthreads = min(cores, cap_threads) if cap_threads else cores
    P = cfg["padlen"]
    ref = O.load_reference()
    if ref is not None:
        kind = "reference"
        tok = ref.Tokenizer(cfg["key"], bool(cfg["eos"]), bool(cfg["bos"]), bool(cfg["padchar"]))
    else:
        kind = "port"
        O.build()
        tok = O.OracleTokenizer(cfg["key"], cfg["eos"], cfg["bos"], cfg["padchar"])

    def run(c, o, nt):
        seqs = synth.unpack(c, o) if ref is not None else None  # the reference's API takes Python objects: untimed
        t0 = time.perf_counter()
        if ref is not None:
            if op == "onehot":
                out = tok.batch_onehot_encode(seqs, padlen=P, destchar=destchar, nthreads=nt)
            else:
                out = tok.batch_tokenize(seqs, padlen=P, destchar=destchar, batch_first=batch_first, nthreads=nt)
        else:
            if op == "onehot":
                out = tok.onehot_packed(c, o, P, destchar, nt)
            else:
                out = tok.tokenize_packed(c, o, P, destchar, batch_first, nt)
        dt = time.perf_counter() - t0
        return dt, out.nbytes

    B = len(offsets) - 1
    total = int(offsets[-1])
    dt, nbytes = run(chars, offsets, nthreads)
    B1 = max(1, B // 8)
    c1, o1 = chars[:int(offsets[B1])], offsets[:B1 + 1]
    dt1, nbytes1 = run(c1, o1, 1)
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "")
    except OSError:
        pass
    return {"cpu_model": model, "value": total / dt / 1e9, "unit": "Gseq-chars/s", "cores": nthreads, "kind": kind,
            "gb_per_s_written": nbytes / dt / 1e9, "seconds": dt, "host_cpus": cores,
            "opt": "-O3 without -march=native (oracle/Makefile); the reference itself ships -O0 -march=native (setup.py:50-55)",
            "sample": "the full batch of this workload (%d sequences, %d chars, %.2f GB output), one call incl. "
                      "result allocation as the reference does per call, nthreads=%d"
                      % (B, total, nbytes / 1e9, nthreads),
            "single_thread": {"value": int(o1[-1]) / dt1 / 1e9, "unit": "Gseq-chars/s", "cores": 1,
                              "gb_per_s_written": nbytes1 / dt1 / 1e9, "seconds": dt1,
                              "sample": "the first %d sequences of the batch (%d chars, %.2f GB output), nthreads=1"
                                        % (B1, int(o1[-1]), nbytes1 / 1e9)}}


def e2e_python_surface(cfg, op, destchar, batch_first, chars, offsets, dev):
    """PCIe-inclusive timings of the DROP-IN surface (bioseq.Tokenizer's Python API) on the same batch -- reported beside the
    kernel numbers, never part of `value`: list[bytes] -> device tensor (every call synchronised, and 20 calls back to back),
    list[bytes] -> numpy (the reference's default return type: includes the D2H copy of the result), packed batch resident
    in HBM -> device tensor including the allocation of the result.  Median of 10 (numpy return: of 5)."""
    import torch
    import bioseq_amd
    from bioseq_amd import synth
    P = cfg["padlen"]
    tok = bioseq_amd.Tokenizer(cfg["key"], bool(cfg["eos"]), bool(cfg["bos"]), bool(cfg["padchar"]))
    seqs = synth.unpack(chars, offsets)
    nthreads = min(8, os.cpu_count() or 1)
    d_chars, d_offs = torch.from_numpy(chars).to(dev), torch.from_numpy(offsets).to(dev)

    def call(device, nt=nthreads):
        if op == "onehot":
            return tok.batch_onehot_encode(seqs, padlen=P, destchar=destchar, nthreads=nt, device=device)
        return tok.batch_tokenize(seqs, padlen=P, destchar=destchar, batch_first=batch_first, nthreads=nt, device=device)

    def packed():
        if op == "onehot":
            return tok.onehot_packed(d_chars, d_offs, P, destchar)
        return tok.tokenize_packed(d_chars, d_offs, P, destchar, batch_first)

    def median_ms(fn, n):
        r = fn()
        del r
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            r = fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            del r
        return float(np.median(ts) * 1e3)

    def pipelined_ms(fn, n=20):
        r = fn()
        del r
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            r = fn()
            del r
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    out = {"nthreads": nthreads, "sequences": len(seqs), "host_cpus": os.cpu_count(),
           "list_to_device_sync_ms": median_ms(lambda: call(dev), 10),
           "list_to_device_sync_default_nthreads_ms": median_ms(lambda: (tok.batch_onehot_encode(seqs, padlen=P, destchar=destchar, device=dev) if op == "onehot" else tok.batch_tokenize(seqs, padlen=P, destchar=destchar, batch_first=batch_first, device=dev)), 10),
           "list_to_device_pipelined20_ms": float(np.median([pipelined_ms(lambda: call(dev)) for _ in range(3)])),
           "packed_resident_to_device_incl_alloc_ms": median_ms(packed, 10),
           "list_to_numpy_ms": median_ms(lambda: call(None), 5),
           "note": "PCIe-inclusive, synthetic list of %d bytes objects; never part of `value`" % len(seqs)}
    return out


def self_launch(n):
    """`python bench.py --gpus N` without a launcher around it: start
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same args>`
    as a child process and relay its output (rank 0's JSON line goes to stdout as it is).  Returns the child's exit code."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: launching %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)  # the first ~10 launches after idle run 3-8 % slow (clock ramp)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive timings of the Python surface (N = 1 only)")
    ap.add_argument("--no-sustained", action="store_true", help="skip the >= 1 s sustained loop (N = 1 only)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="cap the CPU baseline's thread count")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="weak: every rank encodes a batch of the workload's size (default); strong: ONE batch of that "
                         "size is split over the ranks by sequence (sharding.shard_bounds)")
    ap.add_argument("--gather", type=int, default=0, metavar="K",
                    help="N > 1 only: additionally time K whole-batch assemblies over xGMI in every form (all_gather + "
                         "concatenate, grouped point-to-point straight into the destination, to a root and to every "
                         "rank, token matrices + local expansion); reported separately as `gather`, never part of `value`")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # Plain `python bench.py --gpus N`: become the launcher.  Nothing in this process has touched HIP or imported
        # torch yet, and the ranks are CHILD processes (never an exec): torch.distributed.run starts one rank per GPU,
        # rank 0 prints the one JSON line, which is relayed unchanged; exit code = the launcher's.
        sys.exit(self_launch(args.gpus))
    if world != args.gpus:
        args.gpus = world

    import ctypes
    import torch
    import bioseq_amd
    from bioseq_amd import capi, synth

    if not torch.cuda.is_available() or bioseq_amd.device_count() < 1:
        sys.exit("bench.py needs a HIP device: the product has no CPU path")
    # BSQ_BENCH_BACKEND=gloo + BSQ_BENCH_SHARE_GPU=1: smoke-test the N > 1 code path on a 1-GPU box
    # (ranks share device 0; never a measurement).  The driver's runs use nccl = RCCL, one GPU per rank.
    backend = os.environ.get("BSQ_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if os.environ.get("BSQ_BENCH_SHARE_GPU") else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    red_dev = dev if backend == "nccl" else torch.device("cpu")  # where the tiny timing reductions live
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL
        else:
            dist.init_process_group(backend)

    cfg_name, op, destchar, batch_first = WORKLOADS[args.workload]
    cfg = synth.CONFIGS[cfg_name]
    P = cfg["padlen"]
    if args.scaling == "strong":
        # ONE batch of cfg["n"] sequences, rank r owns the contiguous range shard_bounds gives it
        from bioseq_amd.sharding import shard_bounds
        n_job = cfg["n"]
        first, stop = shard_bounds(n_job, world, rank)
        n = stop - first
    else:
        # rank r owns sequences [r*n, (r+1)*n) of the N*n-sequence stream (weak scaling)
        n = cfg["n"]
        n_job = n * world
        first = rank * n
    chars, offsets = synth.synth_packed(cfg["seed"], n, cfg["lo"], cfg["hi"], cfg["letters"], first=first)
    total = int(offsets[-1])

    lib = capi.load()
    desc = capi.make_desc(cfg["key"], cfg["eos"], cfg["bos"], cfg["padchar"])
    C = lib.bsq_alphabet_size(ctypes.byref(desc))
    dt_code = ctypes.c_int(0)
    capi.check(lib.bsq_dtype_from_destchar(destchar.encode(), ctypes.byref(dt_code)))
    sz = lib.bsq_dtype_size(dt_code)
    tdt = {0: torch.int8, 1: torch.int16, 2: torch.int32, 3: torch.int64, 4: torch.float32, 5: torch.float64}[dt_code.value]

    d_chars = torch.from_numpy(chars).to(dev)
    d_offs = torch.from_numpy(offsets).to(dev)
    if op in ("onehot", "onehot_bcl"):
        out = torch.empty((P, n, C) if op == "onehot" else (n, C, P), dtype=tdt, device=dev)
        out_bytes = P * n * C * sz
    else:
        out = torch.empty((n, P) if batch_first else (P, n), dtype=tdt, device=dev)
        out_bytes = P * n * sz
    algo_bytes = total + 8 * (n + 1) + out_bytes  # chars + offsets read once, output written once

    stream = torch.cuda.current_stream()
    sh = ctypes.c_void_p(stream.cuda_stream)

    aug_seed = [0]

    def step():
        if op == "augment+tokenize":  # AugmentedSeqDataset defaults (loaders.py:117-119): chain_len 1, frac 0.5
            aug_seed[0] += 1
            # one C-ABI call = bsq_augment_device, then bsq_tokenize_device (one launch where the fast token kernel applies)
            capi.check(lib.bsq_augment_tokenize_device(ctypes.byref(desc), d_chars.data_ptr(), d_offs.data_ptr(), n, P,
                                                       int(batch_first), dt_code, out.data_ptr(), 1, 0.5, aug_seed[0], sh))
            return
        if op == "onehot":
            st = lib.bsq_onehot_device(ctypes.byref(desc), d_chars.data_ptr(), d_offs.data_ptr(), None, n, P,
                                       dt_code, out.data_ptr(), sh)
        elif op == "onehot_bcl":
            st = lib.bsq_onehot_bcl_device(ctypes.byref(desc), d_chars.data_ptr(), d_offs.data_ptr(), None, n, P,
                                           dt_code, out.data_ptr(), sh)
        else:
            st = lib.bsq_tokenize_device(ctypes.byref(desc), d_chars.data_ptr(), d_offs.data_ptr(), n, P,
                                         int(batch_first), dt_code, out.data_ptr(), sh)
        if st:
            capi.check(st)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed sanity: validate lengths, run once, check a size-independent property
    bad = ctypes.c_int64(-1)
    capi.check(lib.bsq_validate_lengths_device(d_offs.data_ptr(), n, P, desc.bos, desc.eos, ctypes.byref(bad), sh))
    out.fill_(7)
    step()
    torch.cuda.synchronize()
    if op in ("onehot", "onehot_bcl"):
        ones = int(out.sum(dtype=torch.float64).item())
        expect = total + (n if desc.bos else 0) + (n if desc.eos else 0)
        if desc.padchar:
            expect = P * n
        if not os.environ.get("BSQ_BENCH_SKIP_SANITY"):
            assert ones == expect, ("one-hot sanity failed", ones, expect)
    if op == "augment+tokenize":
        # the step mutated d_chars and wrote the tokens of the MUTATED batch (one launch): a plain tokenise of what is in d_chars
        # now must give the same matrix, and about half of the sequences must differ from the pristine batch in one residue
        check = torch.empty_like(out)
        capi.check(lib.bsq_tokenize_device(ctypes.byref(desc), d_chars.data_ptr(), d_offs.data_ptr(), n, P, int(batch_first), dt_code,
                                           check.data_ptr(), sh))
        torch.cuda.synchronize()
        assert torch.equal(check, out), "augment+tokenize sanity failed: tokens are not those of the mutated characters"
        changed = int((d_chars != torch.from_numpy(chars).to(d_chars.device)).sum().item())
        assert 0.4 * n < changed < 0.6 * n, ("augment+tokenize sanity failed: mutations", changed, n)
        del check

    # write-bandwidth yardstick: a plain fill (one 1-KiB store per wave, one aligned 4-KiB chunk per workgroup, blocks in
    # address order) over the same output buffer, in its BEST-KNOWN configuration: 3 resident workgroups per CU (unused
    # LDS as the cap; profiles/r01/fill_occupancy.txt: 6.84 TB/s uncapped, 7.38 at 3 per CU), and uncapped for reference.
    # Measured BEFORE the warm-up steps: its launches also lift the clocks out of idle.
    capi.check(lib.bsq_tuning_set(b"fill_mode", 1))
    fill_bytes = (out_bytes // 16) * 16

    def timed_loop(fn, n, warm):
        for _ in range(warm):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(n):
            fn()
        b.record(stream)
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n  # ms

    fill_gbps = {}
    for name, pad in (("uncapped", 0), ("3_workgroups_per_cu", 53000)):
        capi.check(lib.bsq_tuning_set(b"fill_pad", pad))
        fill_gbps[name] = fill_bytes / (timed_loop(lambda: capi.check(lib.bsq_fill_device(out.data_ptr(), fill_bytes, 0, sh)), 5, 10) * 1e-3) / 1e9
    capi.check(lib.bsq_tuning_set(b"fill_pad", 0))
    fill_best = max(fill_gbps.values())
    # read + write yardstick of the token workloads: the kernel's stream shape (one wave = one aligned 4-KiB chunk of the
    # output + its share of the characters, two dependent load steps like offsets -> characters) with none of its work
    mix_gbps = None
    if op in ("tokenize", "augment+tokenize") and batch_first and sz == 1 and out_bytes % 4096 == 0 and total >= 16:
        src_bytes = (total // 16) * 16
        mix_ms = timed_loop(lambda: capi.check(lib.bsq_copy_mix_device(out.data_ptr(), out_bytes, d_chars.data_ptr(), src_bytes, 1, 1, sh)), 20, 10)
        mix_gbps = (src_bytes + out_bytes) / (mix_ms * 1e-3) / 1e9
    # what one event pair around ONE tiny launch reads: the floor under every per-step event time below
    tiny = torch.empty(4096, dtype=torch.uint8, device=dev)
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in pairs:
        a.record(stream)
        capi.check(lib.bsq_fill_device(tiny.data_ptr(), 4096, 0, sh))
        b.record(stream)
    torch.cuda.synchronize()
    event_floor_ms = float(np.median([a.elapsed_time(b) for a, b in pairs]))

    # Short steps (the 17-55 us token workloads): --warmup W of them is less than a millisecond of GPU time, not enough to lift
    # the clocks out of idle after the host-side pauses above; run the step for ~30 ms first (untimed, like the yardsticks).
    est_ms = timed_loop(step, 5, 2)
    if est_ms < 1.0:
        for _ in range(min(20000, int(30.0 / max(est_ms, 1e-3)))):
            step()
    for _ in range(args.warmup):
        step()
    # THE timed region: exactly K steps between barriers.  One pair of HIP events around the same K launches (recorded on
    # the launch stream) gives the kernel time of a step for the roofline -- rocprofv3's average kernel durations add up
    # to it within 0-7 % on every workload.
    la, lb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    la.record(stream)
    for _ in range(args.steps):
        step()
    lb.record(stream)
    barrier()
    wall = time.perf_counter() - t0
    loop_ms = la.elapsed_time(lb) / args.steps

    # Afterwards, untimed: the same K steps once more with one event pair PER STEP (spread: min / median).  The event
    # records cost ~2 us of queue time per step -- 10 % of the 17-35 us steps of cfg2 / cfg5, nothing on the others --
    # which is why they are not in the timed region.
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a, b in ev:
        a.record(stream)
        step()
        b.record(stream)
    torch.cuda.synchronize()
    kern_ms = [a.elapsed_time(b) for a, b in ev]
    kern_avg_ms = float(np.mean(kern_ms))
    if os.environ.get("BSQ_BENCH_DUMP"):  # per-step device times, for variance hunting
        print("per-step ms:", " ".join("%.3f" % v for v in kern_ms), file=sys.stderr)

    # Sustained: the same step looped for at least one second of wall clock (clocks, thermals), untimed for `value`.
    sustained = None
    if world == 1 and not args.no_sustained:
        n_sus = max(args.steps, int(1.05 / max(loop_ms * 1e-3, 1e-6)))
        sa, sb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        w0 = time.perf_counter()
        # augmentation mutates the batch in place: looped 20 000 times on ONE buffer the sequences drift to the chain's stationary
        # composition (residues with a high self-probability), rejections multiply and the kernel measures that drift, not a fresh
        # batch.  The pristine characters are copied back every 64 steps, INSIDE the timed region (~0.5 us per step).
        pristine = d_chars.clone() if op == "augment+tokenize" else None
        sa.record(stream)
        for it in range(n_sus):
            if pristine is not None and it % 64 == 63:
                d_chars.copy_(pristine)
            step()
        sb.record(stream)
        torch.cuda.synchronize()
        sus_wall = time.perf_counter() - w0
        sus_ms = sa.elapsed_time(sb) / n_sus
        sustained = {"steps": n_sus, "wall_s": sus_wall, "ms_per_step": sus_wall / n_sus * 1e3, "kernel_avg_ms": sus_ms,
                     "frac": algo_bytes / (sus_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "frac_wall": algo_bytes / (sus_wall / n_sus) / 1e9 / HBM_PEAK_GBPS}
        if pristine is not None:
            sustained["restore_every"] = 64  # the batch is reset to its pristine characters every 64 steps, inside the timed region
            del pristine

    gather_info = None
    if world > 1 and args.gather > 0:
        from bioseq_amd import sharding
        seq_first = op == "onehot" or (op == "tokenize" and not batch_first)
        axis = 1 if seq_first else 0
        out_c = out.contiguous() if backend == "nccl" else out.cpu().contiguous()  # gloo smoke runs move host tensors

        def timed(fn, check_axis=None):
            ms = []
            for _ in range(args.gather + 1):
                barrier()
                g0 = time.perf_counter()
                full = fn()
                barrier()
                ms.append((time.perf_counter() - g0) * 1e3)
                if full is not None and check_axis is not None:
                    assert full.shape[check_axis] == n_job, (tuple(full.shape), n_job)
                del full
            t = torch.tensor([float(np.mean(ms[1:]))], dtype=torch.float64, device=red_dev)  # first one warms RCCL up
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        shard_bytes = torch.tensor([float(out_bytes)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(shard_bytes, op=dist.ReduceOp.SUM)
        recv_bytes = float(shard_bytes.item()) - out_bytes
        forms = {}
        forms["all_gather"] = {"ms": timed(lambda: sharding._gather(out_c, axis, n_job), axis),
                               "what": "all_gather of every rank's shard (RCCL picks the algorithm) + concatenation along the "
                                       "batch axis for seq-first layouts; whole batch on every rank"}
        forms["direct_all"] = {"ms": timed(lambda: sharding.gather_direct(out_c, axis, n_job, None), axis),
                               "what": "grouped isend/irecv, one peer per xGMI link, received straight into the destination "
                                       "(one message per peer, or per peer and position row for seq-first layouts); whole "
                                       "batch on every rank"}
        forms["direct_root"] = {"ms": timed(lambda: sharding.gather_direct(out_c, axis, n_job, 0), None),
                                "what": "the same to rank 0 only (the gather north_star names)"}
        if op == "onehot" and backend == "nccl":
            # the xGMI-friendly form: only the uint8 token matrices travel, every rank expands the whole batch itself
            import bioseq_amd as _pkg
            tokz = _pkg.Tokenizer(cfg["key"], cfg["eos"], cfg["bos"], cfg["padchar"])
            raw_tokens, expand = sharding.device_passes(tokz, P, destchar, dev)
            forms["via_tokens"] = {
                "ms": timed(lambda: expand(sharding.gather_direct(raw_tokens(d_chars, d_offs), 1, n_job, None).contiguous()), 1),
                "what": "token pass on the shard + point-to-point gather of the (P, B_g) uint8 token matrices (%d bytes "
                        "received per rank) + local expansion of the whole batch" % int(recv_bytes / max(1, C * sz))}
        if (op == "tokenize" and batch_first) or op in ("onehot_bcl", "onehot"):
            # SURVEY 8e option 3: the encode kernels of every rank store straight into rank 0's buffer (IPC-mapped memory over
            # xGMI): encode AND gather in one step, no collective on the data path.  (The seq-first one-hot goes through the
            # tiled kernel with the root tensor's row pitch, bsq_onehot_block_device.)
            import bioseq_amd as _pkg
            tokz = _pkg.Tokenizer(cfg["key"], cfg["eos"], cfg["bos"], cfg["padchar"])
            try:
                forms["store_into_root"] = {
                    "ms": timed(lambda: sharding.store_shard_into_root(tokz, d_chars, d_offs, first, n_job, P,
                                                                       destchar, {"tokenize": "tokens_bf", "onehot_bcl": "bcl", "onehot": "tbc"}[op], dev, 0, None, False), None),
                    "what": "every rank ENCODES its shard directly into rank 0's buffer through peer-mapped memory "
                            "(sharding.store_shard_into_root): the time includes the encode; no data-path collective"}
            except Exception as ex:  # an IPC / peer-mapping failure must not cost the run its other numbers
                forms["store_into_root"] = {"ms": float("nan"), "what": "failed: %r" % (ex,)}
        for f in forms.values():
            f["gb_per_s_into_each_rank"] = recv_bytes / (f["ms"] * 1e-3) / 1e9
        gather_info = {"bytes_received_per_rank": recv_bytes, "forms": forms, "note": "encode time excluded; mean of %d "
                       "assemblies after one warm-up, MAX over ranks" % args.gather}

    wall_t = torch.tensor([wall], dtype=torch.float64, device=red_dev)
    tot_t = torch.tensor([float(total), float(out_bytes)], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot_t, op=dist.ReduceOp.SUM)
    wall_max = float(wall_t.item())
    job_chars, job_out_bytes = float(tot_t[0].item()), float(tot_t[1].item())

    if rank == 0:
        # kernel time of one step = one pair of events around the K timed launches, / K
        achieved = algo_bytes / (loop_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(args.workload, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        kernel_name = (lib.bsq_onehot_kernel_name(ctypes.byref(desc), n, P, dt_code).decode() if op == "onehot"
                       else (("k_tokens_bp8" if sz == 1 and P >= 128 and P % 16 == 0 else "k_tokenize_chunks")
                             if batch_first else ("k_tokens_pb8_fast" if sz <= 2 and (n * sz) % 16 == 0 and os.environ.get("BSQ_TOKENS_PB8", "0") != "1"
                                                 else ("k_tokens_raw<value>" if sz == 1 else "k_tokenize_tile"))))
        if op == "augment+tokenize":
            kernel_name = ("k_augment_tokens_fused(k_augment_groups -> k_tokens_bp8_fast)" if os.environ.get("BSQ_AUGMENT_FUSED", "0") != "1"
                           else "k_augment_groups+" + kernel_name)
        if op == "onehot_bcl":
            kernel_name = "k_tokens_bp8<raw>+k_expand_bcl" if (P >= 128 and P % 16 == 0 and out_bytes >= (256 << 20)) else "k_tokenize_chunks<onehot bcl>"
        res = {
            "metric": baseline_metric() if args.workload == "cfg3"
                      else "Gseq-chars/s + GB/s written (%s)" % args.workload,
            "value": job_chars * args.steps / wall_max / 1e9,
            "unit": "Gseq-chars/s",
            "gb_per_s_written": job_out_bytes * args.steps / wall_max / 1e9,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall_max / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": {"f": "f32", "B": "u8"}.get(destchar, destchar), "data": "synthetic",
            "config": {"workload": "%s: %s %s, %d seqs/GPU len~U(%d,%d), padlen %d, C=%d, %s output %s" % (
                args.workload, cfg["key"], {"onehot": "batch_onehot_encode", "tokenize": "batch_tokenize", "onehot_bcl": "batch_onehot_encode(layout=bcl)"}.get(op, "BLOSUM62 augment + batch_tokenize"), n,
                cfg["lo"], cfg["hi"], P, C, str(tdt).replace("torch.", ""),
                "(P,B,C)" if op == "onehot" else "(B,C,P)" if op == "onehot_bcl" else ("(B,P)" if batch_first else "(P,B)")),
                "sequences_per_gpu": n, "padlen": P, "channels": C, "input_chars_per_gpu": total,
                "output_bytes_per_gpu": out_bytes,
                "sharding": ("by sequence, no collective; weak scaling: every rank encodes its own %d-sequence batch" % n)
                            if args.scaling == "weak" else
                            ("by sequence, no collective; strong scaling: ONE %d-sequence batch split over %d ranks "
                             "(sharding.shard_bounds; rank 0 holds %d)" % (n_job, world, n)),
                "job_sequences": n_job,
                "rccl_world_size": dist.get_world_size() if world > 1 else 1,
                "backend": (backend + (" (RCCL)" if backend == "nccl" else "")) if world > 1 else "none (single process)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS,
                         "frac_wall": algo_bytes / (wall_max / args.steps) / 1e9 / HBM_PEAK_GBPS,  # from ms_per_step (host clock)
                         "traffic": traffic, "kernel": kernel_name,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "kernel_avg_ms": loop_ms,                    # the K timed steps between ONE pair of events, / K
                         "kernel_avg_ms_per_step_events": kern_avg_ms,  # second, untimed pass: one event pair per step
                         "frac_per_step_events": algo_bytes / (kern_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "kernel_min_ms": float(np.min(kern_ms)), "kernel_median_ms": float(np.median(kern_ms)),
                         "event_pair_floor_ms": event_floor_ms,  # one event pair around one 4-KiB fill: what per-step events add
                         "traffic_source": ("profiles/traffic.json (separate rocprofv3 --pmc passes of this workload, committed; "
                                            "not measured in this run)") if traffic is not None else None,
                         "fill_yardstick_gbps": fill_best, "fill_yardsticks_gbps": fill_gbps,
                         "frac_of_fill": (out_bytes / (loop_ms * 1e-3) / 1e9) / fill_best,
                         "copy_mix_yardstick_gbps": mix_gbps,
                         "frac_of_copy_mix": (achieved / mix_gbps) if mix_gbps else None},
        }
        if sustained is not None:
            res["sustained"] = sustained
        if op == "augment+tokenize":
            # the one-launch form: chunk waves that gave up waiting for their rows' augmentation (expected: 0; then the output is wrong)
            torch.cuda.synchronize()
            nfail = ctypes.c_uint32(0)
            lib.bsq_fused_status(ctypes.byref(nfail))
            res["fused_wait_failures"] = int(nfail.value)
            if res["fused_wait_failures"]:
                raise SystemExit("bsq_augment_tokenize_device: %d token waves gave up waiting" % res["fused_wait_failures"])
        if gather_info is not None:
            res["gather"] = gather_info
        if world == 1 and not args.no_e2e and op in ("onehot", "tokenize"):
            res["e2e"] = e2e_python_surface(cfg, op, destchar, batch_first, chars, offsets, dev)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(cfg, op, destchar, batch_first, chars, offsets, args.cpu_threads or None)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
