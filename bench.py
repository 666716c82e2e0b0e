#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2|cfg2sf|cfg3b|cfg4f|cfg4b|cfg5|cfg5aug]
                    [--configs all|none|a,b,c] [--scaling weak|strong] [--gather K] [--shard-of N [--shard-rank R]] [--cold]

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
  configs       (N = 1, default workload only; --configs) every OTHER BASELINE workload in the same run -- cfg2, cfg2sf, cfg3b,
                cfg4f, cfg4b, cfg5, cfg5aug -- each with ms_per_step / frac (K steps between one pair of events), frac_sustained
                (>= 0.25 s), traffic, kernel, an untimed `check` of its full-size output against the committed folds of the
                REFERENCE's output (tests/golden/bench_folds.json -- never a comparison with the product's own kernels), and
                for the token workloads `cold`: the same step cycling over distinct resident batches whose inputs add up to
                more than 512 MiB (past the 256-MiB Infinity Cache -- a training loop never encodes one batch twice).  Since round 5
                the cold regime's figures ARE the token workloads' `frac` / `ms_per_step` (their working sets fit the Infinity Cache);
                the loop over one resident batch is beside them as `frac_cache_resident` / `ms_per_step_cache_resident`.
                `cold.two_streams` (token workloads): the same cold batches with two streams taking turns -- consecutive batches are
                independent, and only a caller knows that; a side figure (host clock), never `frac` or `value`.
                `<w>_shard8` (cfg3, cfg4f, cfg4b, cfg5aug): rank 0's sharding.shard_bounds share of the workload's batch split over 8
                ranks -- the per-GPU term of the 1/2/4/8 strong-scaling curve, measured on this one GPU (token and cfg4 shards on fresh
                shards like their full-size workloads, the looped figure beside them) -- as a tensor of its own
                (checked against reference-made folds) and written straight into a whole-batch root tensor (`into_root`: a column
                block at the root's pitch for seq-first layouts; also at the middle rank's offset), with
                predicted_strong_scaling_efficiency = t(full batch) / (8 * t(shard)).  `--shard-of N` does the same for --workload.
  build_id      bsq_build_id() of the loaded library; `roofline.traffic_stale` / `configs.*.traffic_stale` are true when the committed
                counters (profiles/traffic.json) were measured on another build.
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
M64 = (1 << 64) - 1

WORKLOADS = {
    #        synth config, op, destchar, batch_first
    "cfg3": ("cfg3", "onehot", "f", False),
    "cfg3b": ("cfg3", "onehot", "B", False),      # the reference's DEFAULT call batch_onehot_encode(seqs, padlen): destchar 'B' -> int8 (tokenize.cpp:81)
    "cfg3bcl": ("cfg3", "onehot_bcl", "f", False),  # channels-first (B,C,P) written directly (loader layout)
    "cfg1oh": ("cfg1", "onehot", "f", False),     # BASELINE configs[0]'s batch as a one-hot: tiny, for smoke runs of the N > 1 path
    "cfg2": ("cfg2", "tokenize", "B", True),
    "cfg2sf": ("cfg2", "tokenize", "B", False),   # the reference's DEFAULT layout of batch_tokenize: (padlen, batch)
    "cfg4f": ("cfg4", "onehot", "f", False),
    "cfg4b": ("cfg4", "onehot", "B", False),
    "cfg5": ("cfg5", "tokenize", "B", True),
    "cfg5aug": ("cfg5", "augment+tokenize", "B", True),  # BASELINE config 5: BLOSUM62 augmentation, then SEB8 tokens
}
DEFAULT_CONFIGS = ["cfg2", "cfg2sf", "cfg3b", "cfg4f", "cfg4b", "cfg5", "cfg5aug"]  # beside the headline cfg3, in the driver's line


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
    cfg1_us = None
    if ref is not None:  # BASELINE configs[0] as the reference runs it: 1000 DNA sequences, nthreads = 1 (its default), median of 50 calls
        c1 = synth.CONFIGS["cfg1"]
        ch, of = synth.synth_packed(c1["seed"], c1["n"], c1["lo"], c1["hi"], c1["letters"])
        s1 = synth.unpack(ch, of, as_str=True)
        t1 = ref.Tokenizer(c1["key"], bool(c1["eos"]), bool(c1["bos"]), bool(c1["padchar"]))
        ts = []
        for _ in range(55):
            t0 = time.perf_counter()
            t1.batch_tokenize(s1, padlen=c1["padlen"], batch_first=True, nthreads=1)
            ts.append(time.perf_counter() - t0)
        cfg1_us = float(np.median(ts[5:]) * 1e6)
    return {"cpu_model": model, "value": total / dt / 1e9, "unit": "Gseq-chars/s", "cores": nthreads, "kind": kind, "cfg1_call_us": cfg1_us,
            "gb_per_s_written": nbytes / dt / 1e9, "seconds": dt, "host_cpus": cores,
            "opt": "-O3 without -march=native (oracle/Makefile); the reference itself ships -O0 -march=native (setup.py:50-55)",
            "sample": "the full batch of this workload (%d sequences, %d chars, %.2f GB output), one call incl. "
                      "result allocation as the reference does per call, nthreads=%d"
                      % (B, total, nbytes / 1e9, nthreads),
            "single_thread": {"value": int(o1[-1]) / dt1 / 1e9, "unit": "Gseq-chars/s", "cores": 1,
                              "gb_per_s_written": nbytes1 / dt1 / 1e9, "seconds": dt1,
                              "sample": "the first %d sequences of the batch (%d chars, %.2f GB output), nthreads=1"
                                        % (B1, int(o1[-1]), nbytes1 / 1e9)}}


def e2e_python_surface(cfg, op, destchar, batch_first, chars, offsets, dev, ndev=2):
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

    def whole_batch_ms():  # the path of rounds 1-3 (scan, pack, one upload, one encode) on the same box, for comparison
        from bioseq_amd import capi
        lib = capi.load()
        capi.check(lib.bsq_tuning_set(b"host_pieces", 1))
        try:
            return median_ms(lambda: call(dev), 10)
        finally:
            capi.check(lib.bsq_tuning_set(b"host_pieces", 0))

    def cfg1_call_us():
        """BASELINE configs[0] -- the reference's own CPU-runnable case: pbeos_tokenizers['DNA'].batch_tokenize on 1000 sequences of
        len <= 256, padlen 256, batch_first -- as a latency: one call -> numpy (the reference's return), one call -> device + sync.
        A call this small is fixed costs; `cpu_baseline.cfg1_call_us` is the reference's own C++ on this box (nthreads = 1)."""
        c1 = synth.CONFIGS["cfg1"]
        ch, of = synth.synth_packed(c1["seed"], c1["n"], c1["lo"], c1["hi"], c1["letters"])
        s1 = synth.unpack(ch, of, as_str=True)
        t1 = bioseq_amd.Tokenizer(c1["key"], bool(c1["eos"]), bool(c1["bos"]), bool(c1["padchar"]))
        return {"tokens_to_numpy_us": median_ms(lambda: t1.batch_tokenize(s1, padlen=c1["padlen"], batch_first=True), 200) * 1e3,
                "tokens_to_device_sync_us": median_ms(lambda: t1.batch_tokenize(s1, padlen=c1["padlen"], batch_first=True, device=dev), 200) * 1e3}

    def on_devices():
        """ONE process, `ndev` devices (sharding.encode_on_devices: the host packs once into pinned memory, device g receives its slice on
        its own copy stream and encodes it on its own stream).  With fewer GPUs visible than `ndev` the entries repeat (stream pairs of one
        GPU: the code path, not N PCIe links)."""
        from bioseq_amd import sharding
        have = torch.cuda.device_count()
        devs = ["cuda:%d" % ((dev.index + g) % have) for g in range(ndev)]

        def shards():
            return sharding.encode_on_devices(tok, seqs, P, destchar, devices=devs, op=op, batch_first=batch_first, nthreads=nthreads)

        def rooted():
            return sharding.encode_on_devices(tok, seqs, P, destchar, devices=devs, op=op, batch_first=batch_first, nthreads=nthreads, root=devs[0])

        def sync_all(fn, n):
            r = fn()
            del r
            for d in range(have):
                torch.cuda.synchronize(d)
            ts = []
            for _ in range(n):
                t0 = time.perf_counter()
                r = fn()
                for d in range(have):
                    torch.cuda.synchronize(d)
                ts.append(time.perf_counter() - t0)
                del r
            return float(np.median(ts) * 1e3)

        return {"devices": devs, "distinct_gpus": len(set(devs)), "shards_sync_ms": sync_all(shards, 7), "into_root_sync_ms": sync_all(rooted, 7),
                "what": "list -> N per-device shards (what parallel_apply consumes) / -> one whole-batch tensor on the first device, all devices synchronised"}

    out = {"nthreads": nthreads, "sequences": len(seqs), "host_cpus": os.cpu_count(), "cfg1_call": cfg1_call_us(),
           "list_to_devices": on_devices(),
           "list_to_device_sync_ms": median_ms(lambda: call(dev), 10),
           "list_to_device_sync_one_upload_one_encode_ms": whole_batch_ms(),
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


def fold_device(t):
    """(xor, sum, wsum) of a device tensor's bytes as tests/golden/make_bench_folds.py defines them, computed ON the device
    with torch integer arithmetic (int64 wraps like uint64); returns three hex strings."""
    import torch
    b = t.contiguous().view(torch.uint8).reshape(-1)
    if b.numel() % 8:
        b = torch.cat([b, torch.zeros(8 - b.numel() % 8, dtype=torch.uint8, device=b.device)])
    w = b.view(torch.int64)
    x = s = ws = 0
    step = 1 << 26
    for i in range(0, w.numel(), step):
        c = w[i:i + step]
        s = (s + int(c.sum().item())) & M64
        k = torch.arange(i, i + c.numel(), dtype=torch.int64, device=c.device) * 2 + 1
        ws = (ws + int((c * k).sum().item())) & M64
        del k
        y = c
        carry = torch.zeros(1, dtype=torch.int64, device=c.device)
        while y.numel() > 1:  # xor of all words by halving
            h = y.numel() // 2
            if y.numel() & 1:
                carry = carry ^ y[-1:]
            y = y[:h] ^ y[h:2 * h]
        x ^= int((y ^ carry).item()) & M64 if y.numel() else int(carry.item()) & M64
    return "%016x" % (x & M64), "%016x" % s, "%016x" % ws


_FOLDS = None


def golden_folds():
    global _FOLDS
    if _FOLDS is None:
        try:
            with open(os.path.join(ROOT, "tests", "golden", "bench_folds.json")) as f:
                _FOLDS = json.load(f)
        except Exception:
            _FOLDS = {}
    return _FOLDS


class Batch:
    """One workload's batch resident in HBM + its step through the C ABI."""

    def __init__(self, name, lib, dev, stream, n=None, first=0, fold_key=None):
        import ctypes
        import torch
        from bioseq_amd import capi, synth
        self.name, self.lib, self.dev, self.stream = name, lib, dev, stream
        self.cfg_name, self.op, self.destchar, self.batch_first = WORKLOADS[name]
        self.cfg = cfg = synth.CONFIGS[self.cfg_name]
        self.P = cfg["padlen"]
        self.n = cfg["n"] if n is None else n
        self.full_size = self.n == cfg["n"] and first == 0
        self.first = first
        self.fold_key = fold_key  # a sub-batch that has reference-made folds of its own (the `<w>_shardN` entries of bench_folds.json)
        self.chars, self.offsets = synth.synth_packed(cfg["seed"], self.n, cfg["lo"], cfg["hi"], cfg["letters"], first=first)
        self.total = int(self.offsets[-1])
        self.desc = capi.make_desc(cfg["key"], cfg["eos"], cfg["bos"], cfg["padchar"])
        self.C = lib.bsq_alphabet_size(ctypes.byref(self.desc))
        self.dt_code = ctypes.c_int(0)
        capi.check(lib.bsq_dtype_from_destchar(self.destchar.encode(), ctypes.byref(self.dt_code)))
        self.sz = lib.bsq_dtype_size(self.dt_code)
        self.tdt = {0: torch.int8, 1: torch.int16, 2: torch.int32, 3: torch.int64, 4: torch.float32, 5: torch.float64}[self.dt_code.value]
        self.d_chars = torch.from_numpy(self.chars).to(dev)
        self.d_offs = torch.from_numpy(self.offsets).to(dev)
        self.out = torch.empty(self.out_shape(self.n), dtype=self.tdt, device=dev)
        self.out_bytes = self.out.numel() * self.sz
        # chars + offsets read once, output written once
        self.algo_bytes = self.total + 8 * (self.n + 1) + self.out_bytes
        self.sh = ctypes.c_void_p(stream.cuda_stream)
        self.aug_seed = 0
        self.step_calls = 0

    def out_shape(self, n):
        P, C = self.P, self.C
        if self.op == "onehot":
            return (P, n, C)
        if self.op == "onehot_bcl":
            return (n, C, P)
        return (n, P) if self.batch_first else (P, n)

    def run(self, d_chars, d_offs, out, n):
        """one pass of the hot path over (d_chars, d_offs) into `out` through the C ABI"""
        import ctypes
        from bioseq_amd import capi
        lib, desc = self.lib, self.desc
        if self.op == "augment+tokenize":  # AugmentedSeqDataset defaults (loaders.py:117-119): chain_len 1, frac 0.5
            self.aug_seed += 1
            st = lib.bsq_augment_tokenize_device(ctypes.byref(desc), d_chars.data_ptr(), d_offs.data_ptr(), n, self.P, int(self.batch_first),
                                                 self.dt_code, out.data_ptr(), 1, 0.5, self.aug_seed, self.sh)
        elif self.op == "onehot":
            st = lib.bsq_onehot_device(ctypes.byref(desc), d_chars.data_ptr(), d_offs.data_ptr(), None, n, self.P, self.dt_code, out.data_ptr(), self.sh)
        elif self.op == "onehot_bcl":
            st = lib.bsq_onehot_bcl_device(ctypes.byref(desc), d_chars.data_ptr(), d_offs.data_ptr(), None, n, self.P, self.dt_code, out.data_ptr(), self.sh)
        else:
            st = lib.bsq_tokenize_device(ctypes.byref(desc), d_chars.data_ptr(), d_offs.data_ptr(), n, self.P, int(self.batch_first), self.dt_code,
                                         out.data_ptr(), self.sh)
        if st:
            capi.check(st)

    def step(self):
        self.step_calls += 1
        self.run(self.d_chars, self.d_offs, self.out, self.n)

    def run_into_root(self, root, b0, n_full):
        """This batch as sequences [b0, b0 + n) of a job of n_full sequences, written straight into the job's whole-batch tensor
        `root` (what sharding.store_shard_into_root does through a peer mapping): seq-first layouts are COLUMN BLOCKS at the root's
        row pitch (bsq_onehot_block_device / bsq_tokenize_block_device), batch-first ones contiguous slabs of rows."""
        import ctypes
        from bioseq_amd import capi
        lib, desc, n, P, C, sz = self.lib, self.desc, self.n, self.P, self.C, self.sz
        base = root.data_ptr()
        if self.op == "onehot":
            st = lib.bsq_onehot_block_device(ctypes.byref(desc), self.d_chars.data_ptr(), self.d_offs.data_ptr(), None, n, P, self.dt_code,
                                             base + b0 * C * sz, n_full, self.sh)
        elif self.op == "tokenize" and not self.batch_first:
            st = lib.bsq_tokenize_block_device(ctypes.byref(desc), self.d_chars.data_ptr(), self.d_offs.data_ptr(), n, P, self.dt_code,
                                               base + b0 * sz, n_full, self.sh)
        else:  # rows [b0, b0 + n) of a batch-first result are contiguous: the whole-tensor entry point at an offset
            rows = root[b0:b0 + n]
            assert rows.is_contiguous()
            self.run(self.d_chars, self.d_offs, rows, n)
            return
        if st:
            capi.check(st)

    def check(self):
        """Untimed: one step into a buffer filled with 7, then (a) a size-independent property, (b) at full size the folds of the
        output against those of the REFERENCE's output (tests/golden/bench_folds.json; cfg5aug: the reference's tokens of the numpy
        twin's mutated batch, seed 1, and the fold of the mutated characters themselves).  Raises on any mismatch."""
        import ctypes
        import torch
        from bioseq_amd import capi
        lib, desc, n, P = self.lib, self.desc, self.n, self.P
        bad = ctypes.c_int64(-1)
        capi.check(lib.bsq_validate_lengths_device(self.d_offs.data_ptr(), n, P, desc.bos, desc.eos, ctypes.byref(bad), self.sh))
        res = {}
        self.out.fill_(7)
        self.aug_seed = 0
        self.step()
        torch.cuda.synchronize()
        if self.op in ("onehot", "onehot_bcl"):
            ones = int(self.out.sum(dtype=torch.float64).item())
            expect = P * n if desc.padchar else self.total + (n if desc.bos else 0) + (n if desc.eos else 0)
            if not os.environ.get("BSQ_BENCH_SKIP_SANITY"):
                assert ones == expect, ("one-hot sanity failed", ones, expect)
            res["ones"] = ones
        if self.op == "augment+tokenize":
            # the step mutated d_chars (seed 1) and wrote the tokens of the MUTATED batch: the generic token kernel on what is in d_chars
            # now must give the same matrix, and about half of the sequences must differ from the pristine batch in one residue
            chk = torch.empty_like(self.out)
            capi.check(lib.bsq_tokenize_device_generic(ctypes.byref(desc), self.d_chars.data_ptr(), self.d_offs.data_ptr(), n, P,
                                                       int(self.batch_first), self.dt_code, chk.data_ptr(), self.sh))
            torch.cuda.synchronize()
            assert torch.equal(chk, self.out), "augment+tokenize sanity failed: tokens are not those of the mutated characters"
            del chk
            changed = int((self.d_chars != torch.from_numpy(self.chars).to(self.dev)).sum().item())
            assert 0.4 * n < changed < 0.6 * n, ("augment+tokenize sanity failed: mutations", changed, n)
            res["mutated_sequences"] = changed
            capi.check(lib.bsq_fused_status(None))
        g = golden_folds().get(self.fold_key or self.name)
        if g and self.fold_key:
            assert g.get("sequences") == n and self.first == 0, ("not the batch these folds were made from", self.fold_key, n, self.first)
        if (self.full_size or self.fold_key) and g:
            x, sm, ws = fold_device(self.out)
            ok = (x, sm, ws) == (g["xor"], g["sum"], g["wsum"]) and self.out_bytes == g["nbytes"]
            assert ok, ("output differs from the reference's (folds)", self.name, (x, sm, ws), g)
            res.update({"vs": "tests/golden/bench_folds.json (folds of the reference's output %s)" % ("of this shard as a batch of its own" if self.fold_key else "at full size"),
                        "xor": x, "sum": sm, "wsum": ws, "ok": True})
            if self.op == "augment+tokenize":
                gm = g["mutated_chars"]
                xm = fold_device(self.d_chars)
                assert xm == (gm["xor"], gm["sum"], gm["wsum"]), ("mutated characters differ from the numpy twin's (folds)", xm, gm)
                assert res["mutated_sequences"] == g["mutated_sequences"]
                res["mutated_chars_ok"] = True
        else:
            res.update({"vs": "size-independent property only (not the full-size single-rank batch)", "ok": True})
        if self.op == "augment+tokenize":  # back to the pristine batch
            self.d_chars.copy_(torch.from_numpy(self.chars).to(self.dev))
        return res

    def kernel_name(self):
        import ctypes
        n, P, sz, op = self.n, self.P, self.sz, self.op
        if op == "onehot":
            return self.lib.bsq_onehot_kernel_name(ctypes.byref(self.desc), n, P, self.dt_code).decode()
        if op == "onehot_bcl":
            return "k_tokens_bp8<raw>+k_expand_bcl" if (P >= 128 and P % 16 == 0 and self.out_bytes >= (256 << 20)) else "k_tokenize_chunks<onehot bcl>"
        # (the library's own dispatch as a function of the shape: bsq_tokenize_kernel_name -- ADVICE round 5: this used to be re-implemented here)
        return self.lib.bsq_tokenize_kernel_name(ctypes.byref(self.desc), n, P, int(self.batch_first), self.dt_code, 1 if op == "augment+tokenize" else 0).decode()

    def describe(self):
        cfg = self.cfg
        return "%s: %s %s, %d seqs/GPU len~U(%d,%d), padlen %d, C=%d, %s output %s" % (
            self.name, cfg["key"], {"onehot": "batch_onehot_encode", "tokenize": "batch_tokenize",
                                    "onehot_bcl": "batch_onehot_encode(layout=bcl)"}.get(self.op, "BLOSUM62 augment + batch_tokenize"),
            self.n, cfg["lo"], cfg["hi"], self.P, self.C, str(self.tdt).replace("torch.", ""),
            "(P,B,C)" if self.op == "onehot" else "(B,C,P)" if self.op == "onehot_bcl" else ("(B,P)" if self.batch_first else "(P,B)"))


def timed_loop(fn, n, warm, stream):
    import torch
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(n):
        fn()
    b.record(stream)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n  # ms


def ramp(step, stream):
    """Short steps (the 17-55 us token workloads): a few warm-up launches are less than a millisecond of GPU time, not enough to
    lift the clocks out of idle after a host-side pause; run the step for ~30 ms first (untimed)."""
    est_ms = timed_loop(step, 5, 2, stream)
    if est_ms < 1.0:
        for _ in range(min(20000, int(30.0 / max(est_ms, 1e-3)))):
            step()
    return est_ms


def sustained_loop(b, min_s, steps_floor, loop_ms, stream):
    """The same step looped for at least `min_s` seconds of wall clock (clocks, thermals)."""
    import torch
    n_sus = max(steps_floor, int(min_s * 1.05 / max(loop_ms * 1e-3, 1e-6)))
    sa, sb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # augmentation mutates the batch in place: looped 20 000 times on ONE buffer the sequences drift to the chain's stationary
    # composition (residues with a high self-probability), rejections multiply and the kernel measures that drift, not a fresh
    # batch.  The pristine characters are copied back every 64 steps, INSIDE the timed region (~0.5 us per step).
    pristine = b.d_chars.clone() if b.op == "augment+tokenize" else None
    torch.cuda.synchronize()
    w0 = time.perf_counter()
    sa.record(stream)
    for it in range(n_sus):
        if pristine is not None and it % 64 == 63:
            b.d_chars.copy_(pristine)
        b.step()
    sb.record(stream)
    torch.cuda.synchronize()
    sus_wall = time.perf_counter() - w0
    sus_ms = sa.elapsed_time(sb) / n_sus
    out = {"steps": n_sus, "wall_s": sus_wall, "ms_per_step": sus_wall / n_sus * 1e3, "kernel_avg_ms": sus_ms,
           "frac": b.algo_bytes / (sus_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
           "frac_wall": b.algo_bytes / (sus_wall / n_sus) / 1e9 / HBM_PEAK_GBPS}
    if pristine is not None:
        out["restore_every"] = 64  # the batch is reset to its pristine characters every 64 steps, inside the timed region
        b.d_chars.copy_(pristine)
    return out


def cold_regime(b, steps, min_s, stream, verify=True):
    """The token workloads with NOTHING of the batch resident in a cache: the same step cycling over NB distinct batches (inputs and
    outputs in their own buffers) whose inputs add up to more than 512 MiB -- twice the 256-MiB Infinity Cache -- so that every
    character and offset comes from HBM, as in a training loop that never encodes one batch twice (bioseq/loaders.py:76-104).
    Batch k is the workload's batch with its sequences rotated by k * 4099 positions (built on the device); after the measurement
    every output is checked against the rotated output of batch 0 (itself checked against the reference's folds).  The copy-mix
    yardstick (the kernel's stream shape with none of its work) is run the same way."""
    import torch
    from bioseq_amd import capi
    n, dev = b.n, b.dev
    in_bytes = b.total + 8 * (n + 1)
    nb = max(8, -(-(513 << 20) // in_bytes))
    offs = b.d_offs
    batches = []
    for k in range(nb):
        r = (k * 4099) % n
        c0 = int(b.offsets[r])
        ch = torch.cat([b.d_chars[c0:], b.d_chars[:c0]]) if r else b.d_chars.clone()
        lens = offs[1:] - offs[:-1]
        lens = torch.cat([lens[r:], lens[:r]])
        of = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        of[1:] = torch.cumsum(lens, 0)
        batches.append((ch, of, torch.empty_like(b.out), r))
    pristine = [c.clone() for c, _, _, _ in batches] if b.op == "augment+tokenize" else None
    seeds = [0] * nb

    def one(k, it):
        ch, of, out, _ = batches[k]
        b.run(ch, of, out, n)
        if pristine is not None and (it // nb) % 64 == 63:
            ch.copy_(pristine[k])  # right after use: seven other batches pass before it is read again

    it = [0]

    def step():
        one(it[0] % nb, it[0])
        it[0] += 1

    ramp(step, stream)
    loop_ms = timed_loop(step, max(steps, 2 * nb), 2 * nb, stream)
    n_sus = max(steps, int(min_s * 1.05 / max(loop_ms * 1e-3, 1e-6)))
    sus_ms = timed_loop(step, n_sus, 0, stream)
    res = {"batches": nb, "input_bytes_total": nb * in_bytes, "output_bytes_total": nb * b.out_bytes,
           "ms_per_step": loop_ms, "frac": b.algo_bytes / (loop_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
           "sustained_steps": n_sus, "sustained_ms_per_step": sus_ms,
           "frac_sustained": b.algo_bytes / (sus_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    # Beside the in-order figure: the same cold batches with TWO streams taking turns.  Consecutive batches are independent, but only
    # the caller knows that -- on one stream every launch waits for the one before it, so each 16-40-us launch pays its own ramp-up and
    # drain; on two streams the tail of one hides under the head of the next (round 5, scripts/two_stream_lab.py: cfg2 22.1 -> 18.1 us per
    # batch, the cold copy stream's own rate).  Host clock over N launches between two device synchronisations; never `frac`, never
    # `value`; token workloads only (the fused augmentation restores its pristine characters on the main stream); a failure here is
    # recorded, not raised.
    if b.op == "tokenize":
        try:
            import ctypes
            side = torch.cuda.Stream(device=dev)
            handles = [b.sh, ctypes.c_void_p(side.cuda_stream)]

            def run2(nl):
                for i in range(nl):
                    b.sh = handles[i & 1]
                    one(i % nb, i)

            torch.cuda.synchronize()
            run2(4 * nb)
            torch.cuda.synchronize()
            nl = max(1000, 2 * steps)
            t0 = time.perf_counter()
            run2(nl)
            torch.cuda.synchronize()
            two_ms = (time.perf_counter() - t0) / nl * 1e3
            res["two_streams"] = {"ms_per_step": two_ms, "frac": b.algo_bytes / (two_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "launches": nl,
                                  "what": "the same cold batches, two streams taking turns (independent batches; host clock between two "
                                          "synchronisations) -- what a loader that alternates streams gets; not the in-order figure above"}
        except Exception as ex:  # noqa: BLE001 -- a side measurement must never cost the driver its line
            res["two_streams"] = {"error": repr(ex)}
        finally:
            b.sh = handles[0] if "handles" in locals() else b.sh
            torch.cuda.synchronize()
    # The same cold batches FOUR PER LAUNCH (round 6: bsq_tokenize_device_multi -- the grid is the concatenation of four batches' grids,
    # one ramp-up and one drain per launch): what an in-order caller that has its next batches at hand gets on ONE stream.  HIP events
    # on the launch stream; per BATCH; never `frac` or `value` of the workload (those stay the one-batch-per-launch figures).
    multi_out = None
    if b.op == "tokenize" and hasattr(b.lib, "bsq_tokenize_device_multi"):
        import ctypes
        per = 4
        groups = []
        for g in range(nb // per):
            arr = (capi.Batch * per)()
            for j in range(per):
                ch, of, out, _ = batches[g * per + j]
                arr[j].chars, arr[j].offsets, arr[j].B, arr[j].out = ch.data_ptr(), of.data_ptr(), n, out.data_ptr()
            groups.append(arr)
        gi = [0]

        def stepm():
            st = b.lib.bsq_tokenize_device_multi(ctypes.byref(b.desc), per, groups[gi[0] % len(groups)], b.P, int(b.batch_first), b.dt_code, b.sh)
            if st:
                capi.check(st)
            gi[0] += 1

        for _, _, out, _ in batches:
            out.fill_(7)
        for _ in groups:
            stepm()
        torch.cuda.synchronize()
        multi_out = [batches[k][2].clone() for k in range(len(groups) * per)] if verify else None
        m_ms = timed_loop(stepm, max(200, n_sus // per), 4 * len(groups), stream) / per
        res["multi4"] = {"ms_per_step": m_ms, "frac": b.algo_bytes / (m_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "batches_per_launch": per,
                         "what": "bsq_tokenize_device_multi: four of the cold batches per launch on the one in-order stream, per batch"}
    if b.op == "augment+tokenize" and hasattr(b.lib, "bsq_augment_tokenize_device_multi"):
        # BASELINE config 5 with augmentation, FOUR fresh batches per call (round 6: bsq_augment_tokenize_device_multi -- one augmentation
        # launch + one token launch for the four; nobody waits inside a kernel, the characters are read once by plain loads).  Same
        # restore discipline as the one-batch loop above; per BATCH; beside the one-batch-per-call figure, never instead of it.
        import ctypes
        per = 4
        groups = []
        for g in range(nb // per):
            arr = (capi.Batch * per)()
            for j in range(per):
                ch, of, out, _ = batches[g * per + j]
                arr[j].chars, arr[j].offsets, arr[j].B, arr[j].out = ch.data_ptr(), of.data_ptr(), n, out.data_ptr()
            groups.append(arr)
        gi = [0]

        def stepa():
            k = gi[0]
            g = k % len(groups)
            sd = (ctypes.c_uint64 * per)(*[4 * k + j + 1 for j in range(per)])
            st = b.lib.bsq_augment_tokenize_device_multi(ctypes.byref(b.desc), per, groups[g], b.P, int(b.batch_first), b.dt_code, 1, 0.5, sd, b.sh)
            if st:
                capi.check(st)
            if (k // len(groups)) % 64 == 63:
                for j in range(per):
                    batches[g * per + j][0].copy_(pristine[g * per + j])
            gi[0] += 1

        for k in range(nb):
            batches[k][0].copy_(pristine[k])
        a_ms = timed_loop(stepa, max(200, n_sus // per), 4 * len(groups), stream) / per
        for k in range(nb):
            batches[k][0].copy_(pristine[k])
        torch.cuda.synchronize()
        res["multi4"] = {"ms_per_step": a_ms, "frac": b.algo_bytes / (a_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "batches_per_call": per,
                         "what": "bsq_augment_tokenize_device_multi: four of the cold batches per call (one augmentation launch + one token launch), per batch",
                         "check": "tests/test_multi_batch.py: bit-identical to the per-batch calls with the same seeds"}
    # untimed: every batch once more from its pristine characters, compared with batch 0's output rotated
    if not verify:
        res["check"] = "none (a lab run with result-changing ablations)"
    elif b.op != "augment+tokenize":
        for k in range(nb):
            one(k, 0)
        torch.cuda.synchronize()
        base = batches[0][2]
        axis = 0 if b.batch_first else 1
        for k in range(1, nb):
            assert torch.equal(batches[k][2], torch.roll(base, -batches[k][3], dims=axis)), ("cold batch differs", b.name, k)
        res["check"] = "every batch's output == the reference-checked output of batch 0, rotated"
        if multi_out is not None:
            for k, mo in enumerate(multi_out):
                assert torch.equal(mo, batches[k][2]), ("a batch of the multi-batch launch differs from its own launch", b.name, k)
            res["multi4"]["check"] = "every batch of the four-batch launches == its one-batch launch (itself checked against batch 0, rotated)"
        del multi_out
    else:
        capi.check(b.lib.bsq_fused_status(None))
        res["check"] = "bsq_fused_status ok (the mutated batches are checked on batch 0 by `check`)"
    if b.batch_first and b.sz == 1 and b.out_bytes % 4096 == 0 and b.total >= 16:
        src_bytes = (b.total // 16) * 16

        def mix():
            ch, _, out, _ = batches[it[0] % nb]
            capi.check(b.lib.bsq_copy_mix_device(out.data_ptr(), b.out_bytes, ch.data_ptr(), src_bytes, 1, 1, b.sh))
            it[0] += 1

        mix_ms = timed_loop(mix, max(200, 4 * nb), 2 * nb, stream)
        res["copy_mix_ms"] = mix_ms
        res["frac_of_copy_mix"] = mix_ms / sus_ms
    del batches, pristine
    torch.cuda.empty_cache()
    return res


# One-hot workloads whose INPUT (158 MB of reads at cfg4) survives in the 256-MiB Infinity Cache from one step of the loop to the next and
# is a visible share of the step's traffic: cfg4 int8 reads 0.70 of the roof looped over one batch and 0.62 on fresh batches (round 5:
# scripts/size_sweep.py).  Like the token workloads they get the cold regime, and it is their primary figure.
COLD_ONEHOT = ("cfg4f", "cfg4b")
SHARD_CONFIGS = ["cfg3", "cfg4f", "cfg4b", "cfg5aug"]  # their rank-0 shard of an 8-rank strong split rides in the driver's line as `<w>_shard8`


def run_shard(name, world, lib, dev, stream, steps, warmup, full_ms=None, full_cold_ms=None, rank=0):
    """The per-GPU term of the 1/2/4/8 strong-scaling curve, measured on ONE GPU (missing #1): rank `rank`'s
    sharding.shard_bounds share of the workload's batch -- a 1/world-size launch -- (i) as a tensor of its own and (ii) written
    straight into a whole-batch root tensor (column block at the root's pitch for seq-first layouts, slab of rows otherwise; rank 0's
    position and the middle rank's, whose column offset need not be chunk-aligned).  Rank 0's stand-alone output is checked against
    reference-made folds (`<w>_shard8` in tests/golden/bench_folds.json); every block against the stand-alone tensor, and the root
    around it for stray writes.  predicted_strong_scaling_efficiency = t(full batch) / (world * t(shard))."""
    import torch
    from bioseq_amd.sharding import shard_bounds
    from bioseq_amd import synth
    t0 = time.perf_counter()
    cfg_name, op, _, batch_first = WORKLOADS[name]
    n_full = synth.CONFIGS[cfg_name]["n"]
    b0, b1 = shard_bounds(n_full, world, rank)
    key = "%s_shard%d" % (name, world)
    b = Batch(name, lib, dev, stream, n=b1 - b0, first=b0, fold_key=key if (rank == 0 and key in golden_folds()) else None)
    res = {"workload": b.describe(), "shard_of": world, "rank": rank, "sequences": b.n, "job_sequences": n_full,
           "kernel": b.kernel_name(), "algorithmic_bytes_per_launch": b.algo_bytes}
    res["check"] = b.check()
    ramp(b.step, stream)
    loop_ms = timed_loop(b.step, steps, warmup, stream)
    if op == "augment+tokenize":
        b.d_chars.copy_(torch.from_numpy(b.chars).to(dev))
    sus = sustained_loop(b, 0.25, steps, loop_ms, stream)
    res.update({"ms_per_step": loop_ms, "frac": b.algo_bytes / (loop_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "frac_sustained": sus["frac"],
                "sustained_ms_per_step": sus["kernel_avg_ms"]})
    if full_ms:
        res["full_batch_ms_per_step"] = full_ms
        res["predicted_strong_scaling_efficiency"] = full_ms / (world * loop_ms)
    if op in ("tokenize", "augment+tokenize") or name in COLD_ONEHOT:
        # (round 6: the cfg4 one-hot shards too -- like their full-size workloads: a shard's 20 MB of reads stay cache-resident between the steps of a
        #  loop over ONE shard, which a rank of a real job never sees; the looped figure stays beside it)
        cold = cold_regime(b, steps, 0.25, stream)
        res["cold"] = cold
        res["frac_cache_resident"], res["ms_per_step_cache_resident"] = res["frac"], res["ms_per_step"]
        res["frac"], res["ms_per_step"] = cold["frac"], cold["ms_per_step"]
        res["regime"] = "cold (inputs and outputs of > 512 MiB of distinct shards in turn); the cache-resident loop is beside it"
        if full_cold_ms:
            res["full_batch_cold_ms_per_step"] = full_cold_ms
            res["predicted_strong_scaling_efficiency_cold"] = full_cold_ms / (world * cold["ms_per_step"])
    # (ii) into the root's whole-batch tensor
    seq_first = op == "onehot" or (op == "tokenize" and not batch_first)
    root = torch.empty(b.out_shape(n_full), dtype=b.tdt, device=dev)
    b.aug_seed = 0
    b.step()  # the stand-alone tensor once more from the pristine batch (seed 1): what every block must equal
    want = b.out.clone()
    if op == "augment+tokenize":
        b.d_chars.copy_(torch.from_numpy(b.chars).to(dev))
    ranks = [rank] if world < 3 or rank != 0 else [0, world // 2]
    for r in ranks:
        # (timing only needs a position: the middle rank's block is THIS shard's sequences at that rank's column / row offset)
        p0 = shard_bounds(n_full, world, r)[0]
        if p0 + b.n > n_full:
            continue
        root.fill_(7)
        b.aug_seed = 0
        b.run_into_root(root, p0, n_full)
        torch.cuda.synchronize()
        got = root[:, p0:p0 + b.n] if seq_first else root[p0:p0 + b.n]
        assert torch.equal(got, want), ("a block written into the root differs from the stand-alone shard", name, r)
        seven = torch.tensor(7, dtype=b.tdt, device=dev)
        left = root[:, :p0] if seq_first else root[:p0]
        right = root[:, p0 + b.n:] if seq_first else root[p0 + b.n:]
        assert bool((left == seven).all()) and bool((right == seven).all()), ("a block wrote outside its columns / rows", name, r)
        if op == "augment+tokenize":
            b.d_chars.copy_(torch.from_numpy(b.chars).to(dev))
        blk = lambda: b.run_into_root(root, p0, n_full)
        ramp(blk, stream)
        ms = timed_loop(blk, steps, warmup, stream)
        if op == "augment+tokenize":
            b.d_chars.copy_(torch.from_numpy(b.chars).to(dev))
        e = {"ms_per_step": ms, "frac": b.algo_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "first_sequence": p0,
             "byte_offset_in_root_mod_4096": (p0 * (b.C if op == "onehot" else 1) * b.sz if seq_first else p0 * b.out_bytes // b.n) % 4096,
             "what": ("column block of the (P, %d%s) root at its row pitch" % (n_full, ", C" if op == "onehot" else "")) if seq_first
                     else "rows [%d, %d) of the root: a contiguous slab" % (p0, p0 + b.n),
             "check": "== the stand-alone shard tensor; the rest of the root untouched"}
        if full_ms:
            e["predicted_strong_scaling_efficiency"] = full_ms / (world * ms)
        res["into_root" if r == rank else "into_root_at_rank%d" % r] = e
    if op == "augment+tokenize":
        from bioseq_amd import capi
        torch.cuda.synchronize()
        capi.check(lib.bsq_fused_status(None))
    res["seconds"] = time.perf_counter() - t0
    del b, root, want
    torch.cuda.empty_cache()
    return res


def loader_epochs(dev, batch_size=4096, epochs=3, consumer_n=0):
    """The loader layer (reference: DataLoader over FlatFileDataset, bioseq/loaders.py:76-104) on a resident store of
    BASELINE config 5's sequences (262 144 SEB8 sequences, len ~ U(30,512)): one shuffled epoch of `FlatFileDataset.batches` -- device-side
    permutation, gather, (augmentation,) encode; 64 batches of 4096 -- in order on one stream, with `prefetch=2` (the next batches on
    two side streams, round 6) and with `group=4` (four batches gathered + encoded as one super-batch, handed out as views).  Microseconds per batch, host clock over whole epochs (median of `epochs`), the consumer only keeps a
    reference to the batch -- the WORST case for prefetching (the loop is then bound by the host's ~20 us of launches per batch, and
    the hand-off adds its own) --, or (consumer_n > 0) runs a model-sized kernel per batch on its stream (a consumer_n^3 bf16 matmul) that
    the next batch's encode can hide under.  A side figure: never `value`."""
    import tempfile
    import torch
    import bioseq_amd
    from bioseq_amd import synth
    from bioseq_amd.flatfile import FlatFile
    from bioseq_amd.loaders import AugmentedSeqDataset, FlatFileDataset
    c = synth.CONFIGS["cfg5"]
    chars, offs = synth.synth_packed(c["seed"], c["n"], c["lo"], c["hi"], c["letters"])
    res = {"sequences": c["n"], "batch_size": batch_size, "batches_per_epoch": -(-c["n"] // batch_size), "consumer_matmul_n": consumer_n}
    if consumer_n:
        wa = torch.randn(consumer_n, consumer_n, device=dev, dtype=torch.bfloat16)
        wb = torch.randn(consumer_n, consumer_n, device=dev, dtype=torch.bfloat16)
        wc = torch.empty_like(wa)
        for _ in range(3):
            torch.mm(wa, wb, out=wc)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            torch.mm(wa, wb, out=wc)
        torch.cuda.synchronize()
        res["consumer_us"] = (time.perf_counter() - t0) / 50 * 1e6
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "store.ff")
        with open(path, "wb") as f:  # the FlatFile layout (src/fxstats.cpp:33-64): count, offsets, characters
            f.write(np.array([c["n"]], dtype="<u8").tobytes())
            f.write(offs.astype("<u8").tobytes())
            f.write(chars.tobytes())
        ff = FlatFile(path)
        tok = bioseq_amd.Tokenizer(c["key"], bool(c["eos"]), bool(c["bos"]), bool(c["padchar"]))
        kinds = {"tokens_int64": lambda: FlatFileDataset(ff, tok, device=dev),
                 "tokens_int8": lambda: FlatFileDataset(ff, tok, device=dev, token_dtype="b"),
                 "augment_tokens_int8": lambda: AugmentedSeqDataset(ff, tok, device=dev, token_dtype="b"),
                 "onehot_bcl_f32": lambda: FlatFileDataset(ff, tok, device=dev, cnn=True)}
        for name, make in kinds.items():
            ds = make()
            out = {}
            for label, opts in (("prefetch0", {}), ("prefetch2", {"prefetch": 2}), ("group4", {"group": 4})):
                ts = []
                for ep in range(epochs + 1):
                    g = torch.Generator(device=dev).manual_seed(ep)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    last = None
                    for batch in ds.batches(batch_size, shuffle=True, generator=g, **opts):
                        last = batch
                        if consumer_n:
                            torch.mm(wa, wb, out=wc)
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0)
                    del last
                out["%s_us_per_batch" % label] = float(np.median(ts[1:])) / res["batches_per_epoch"] * 1e6
            out["speedup"] = out["prefetch0_us_per_batch"] / out["prefetch2_us_per_batch"]
            res[name] = out
            del ds
        del ff
    torch.cuda.empty_cache()
    return res


def traffic_of(workload):
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            return json.load(open(tpath)).get(workload, {}).get("hbm_bytes_per_launch")
        except Exception:
            return None
    return None


def traffic_build_id(workload):
    """build id (bsq_build_id()) of the library the committed counters of `workload` were measured on, or None"""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        return (t.get(workload) or {}).get("build_id") or t.get("_build_id")
    except Exception:
        return None


def traffic_stale(workload, lib):
    """True when profiles/traffic.json's counters of this workload come from ANOTHER build of the library than the one loaded
    (a kernel change since the last profiling pass is then invisible in `traffic`); None when there are no counters."""
    if traffic_of(workload) is None:
        return None
    return traffic_build_id(workload) != lib.bsq_build_id().decode()


def run_config(name, lib, dev, stream, steps, warmup):
    """One of the OTHER BASELINE workloads inside the default run (N = 1): check, K-step loop, >= 0.25 s sustained, cold regime."""
    import torch
    t0 = time.perf_counter()
    b = Batch(name, lib, dev, stream)
    res = {"workload": b.describe(), "kernel": b.kernel_name(), "algorithmic_bytes_per_launch": b.algo_bytes}
    res["check"] = b.check()
    ramp(b.step, stream)
    loop_ms = timed_loop(b.step, steps, warmup, stream)
    if b.op == "augment+tokenize":
        b.d_chars.copy_(torch.from_numpy(b.chars).to(dev))
    res["ms_per_step"] = loop_ms
    res["frac"] = b.algo_bytes / (loop_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
    sus = sustained_loop(b, 0.25, steps, loop_ms, stream)
    res["frac_sustained"] = sus["frac"]
    res["sustained"] = sus
    res["gseq_chars_per_s"] = b.total / (loop_ms * 1e-3) / 1e9
    res["gb_per_s_written"] = b.out_bytes / (loop_ms * 1e-3) / 1e9
    res["traffic"] = traffic_of(name)
    res["traffic_stale"] = traffic_stale(name, lib)
    if b.op in ("tokenize", "augment+tokenize") or name in COLD_ONEHOT:
        # (COLD_ONEHOT: the cfg4 one-hots, whose 158 MB of input stay cache-resident between the steps of a loop over one batch)
        # The token workloads' working sets (35 + 64 MiB, 71 + 128 MiB) fit the 256-MiB Infinity Cache, and a training loop never
        # encodes one batch twice: their PRIMARY figures are the cold regime's; the loop over one resident
        # batch stays beside them, labelled.
        cold = res["cold"] = cold_regime(b, steps, 0.5, stream)
        res["frac_cache_resident"], res["ms_per_step_cache_resident"] = res["frac"], res["ms_per_step"]
        res["frac_sustained_cache_resident"] = res["frac_sustained"]
        res["frac"], res["ms_per_step"], res["frac_sustained"] = cold["frac"], cold["ms_per_step"], cold["frac_sustained"]
        res["gseq_chars_per_s"] = b.total / (cold["ms_per_step"] * 1e-3) / 1e9
        res["gb_per_s_written"] = b.out_bytes / (cold["ms_per_step"] * 1e-3) / 1e9
        res["regime"] = ("cold: the step cycling over %d distinct batches (%.0f MB of inputs, %.0f MB of outputs), nothing cache-resident; "
                         "`*_cache_resident` = the same step looped over ONE batch" % (cold["batches"], cold["input_bytes_total"] / 1e6, cold["output_bytes_total"] / 1e6))
    if b.op == "augment+tokenize":
        from bioseq_amd import capi
        torch.cuda.synchronize()
        capi.check(lib.bsq_fused_status(None))
    if name in ("cfg2sf", "cfg3b"):
        # the reference's DEFAULT call of this entry point, end to end: a Python list in, a numpy array out (tokenize.cpp:65-98:
        # destchar 'B', batch_first False) -- host scan + pack, H2D, kernel, D2H of the result.  PCIe-inclusive, never a `frac`.
        import bioseq_amd
        from bioseq_amd import synth
        tok = bioseq_amd.Tokenizer(b.cfg["key"], bool(b.cfg["eos"]), bool(b.cfg["bos"]), bool(b.cfg["padchar"]))
        seqs = synth.unpack(b.chars, b.offsets)
        call = (lambda: tok.batch_onehot_encode(seqs, padlen=b.P)) if b.op == "onehot" else (lambda: tok.batch_tokenize(seqs, padlen=b.P))
        r = call()
        assert isinstance(r, np.ndarray) and r.dtype == np.int8 and r.shape == tuple(b.out.shape)
        assert fold_device(torch.from_numpy(r)) == (res["check"]["xor"], res["check"]["sum"], res["check"]["wsum"])
        ts = []
        for _ in range(6):  # (the previous result is released first: a second 1.3-GB array alive during the call doubles the page faults)
            del r
            t1 = time.perf_counter()
            r = call()
            ts.append(time.perf_counter() - t1)
        ts = ts[1:]
        res["e2e_reference_default_call_ms"] = float(np.median(ts) * 1e3)
        res["e2e_reference_default_call"] = ("tok.batch_onehot_encode(seqs, padlen)" if b.op == "onehot" else "tok.batch_tokenize(seqs, padlen)") + \
                                            " -> numpy int8 %r (%.0f MB over PCIe back to the host)" % (tuple(r.shape), r.nbytes / 1e6)
        del r, seqs
    res["seconds"] = time.perf_counter() - t0
    del b
    torch.cuda.empty_cache()
    return res


LINE_LIMIT = 4096  # bytes: the driver's reader lost round 5's 22-KB line (BENCH_r05.json: parsed null); 11.5 KB still parsed in round 4


def _r(v, sig=5):
    """floats to `sig` significant digits (the line is a record, not a transport for doubles); everything else unchanged"""
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (sig, v))
    return v


def compact_line(res):
    """The ONE stdout line: the contract's keys + `roofline` + `cpu_baseline` + `check.ok` + `build_id`, and `configs` reduced to
    {name: {ms, frac, cold, ok}} -- at most LINE_LIMIT bytes whatever was measured.  Everything else of `res` (sustained, e2e, two_streams,
    into_root, yardsticks, prose) goes to bench_full.json (`emit`)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: _r(res[k], 7) for k in keep if k in res}
    cfg = res.get("config") or {}
    line["config"] = {k: cfg[k] for k in ("workload", "sequences_per_gpu", "padlen", "channels", "input_chars_per_gpu", "output_bytes_per_gpu",
                                          "job_sequences", "rccl_world_size", "backend") if k in cfg}
    if "sharding" in cfg:
        line["config"]["sharding"] = cfg["sharding"].split(";")[0][:48]
    rf = res.get("roofline") or {}
    line["roofline"] = {k: _r(rf.get(k), 6) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_stale", "kernel",
                                                      "algorithmic_bytes_per_launch", "kernel_avg_ms") if k in rf}
    for k in ("frac_wall", "frac_of_fill", "frac_of_copy_mix"):
        if rf.get(k) is not None:
            line["roofline"][k] = _r(rf[k], 4)
    cb = res.get("cpu_baseline")
    if cb:
        c = {k: _r(cb.get(k)) for k in ("cpu_model", "value", "unit", "cores", "kind", "seconds") if k in cb}
        c["sample"] = (cb.get("sample") or "")[:100]
        st = cb.get("single_thread")
        if st:
            c["single_thread"] = {"value": _r(st.get("value")), "cores": st.get("cores")}
        line["cpu_baseline"] = c
    line["gb_per_s_written"] = _r(res.get("gb_per_s_written"))
    line["check"] = {"ok": bool((res.get("check") or {}).get("ok"))}
    line["build_id"] = res.get("build_id")
    for k in ("cold", "shard"):
        if isinstance(res.get(k), dict):
            line[k] = {"ms": _r(res[k].get("ms_per_step"), 4), "frac": _r(res[k].get("frac"), 4)}
    if isinstance(res.get("gather"), dict):
        g = res["gather"]
        line["gather"] = {"rccl_world_size": g.get("rccl_world_size"), "rccl_version": g.get("rccl_version"),
                          "ms": {k: _r(v.get("ms"), 4) for k, v in (g.get("forms") or {}).items()}}
    if "fused_wait_failures" in res:
        line["fused_wait_failures"] = res["fused_wait_failures"]
    if isinstance(res.get("loader"), dict) and "error" not in res["loader"]:  # us per batch: [in order, prefetch = 2, group = 4]
        line["loader_us_per_batch"] = {k: [_r(v.get("prefetch0_us_per_batch"), 3), _r(v.get("prefetch2_us_per_batch"), 3), _r(v.get("group4_us_per_batch"), 3)]
                                       for k, v in res["loader"].items() if isinstance(v, dict)}
    if res.get("configs"):
        line["configs"] = {}
        for name, c in res["configs"].items():
            e = {"ms": _r(c.get("ms_per_step"), 4), "frac": _r(c.get("frac"), 3), "cold": "cold" in c, "ok": bool((c.get("check") or {}).get("ok"))}
            m4 = ((c.get("cold") or {}).get("multi4") or {}).get("frac")
            if m4 is not None:
                e["multi4"] = _r(m4, 3)
            line["configs"][name] = e
    line["full"] = "bench_full.json"
    text = json.dumps(line, separators=(",", ":"))
    # whatever a future round adds: the line never outgrows its reader again (drop the optional parts in this order)
    for drop in ("loader_us_per_batch", "gather", "shard", "cold", "configs"):
        if len(text.encode()) <= LINE_LIMIT:
            break
        line.pop(drop, None)
        text = json.dumps(line, separators=(",", ":"))
    assert len(text.encode()) <= LINE_LIMIT, len(text.encode())
    return text


def emit(res, full_line=False):
    """Full result -> bench_full.json beside this script (and under gpurun_out/ when that exists, so it comes back from the GPU box);
    the compact line -> stdout, the ONLY thing this process prints there."""
    blob = json.dumps(res, indent=1)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, os.environ.get("BSQ_BENCH_FULL", "bench_full.json")), "w") as f:
                    f.write(blob + "\n")
            except OSError as ex:
                print("bench.py: could not write bench_full.json into %s: %r" % (d, ex), file=sys.stderr)
    sys.stdout.flush()
    print(json.dumps(res) if full_line else compact_line(res), flush=True)



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)  # the first ~10 launches after idle run 3-8 % slow (clock ramp)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--configs", default=None,
                    help="N = 1: other workloads measured in the same run and reported under `configs`: 'all' (the default for the "
                         "default workload: %s), 'none', or a comma-separated list" % ",".join(DEFAULT_CONFIGS))
    ap.add_argument("--no-configs", action="store_true", help="same as --configs none")
    ap.add_argument("--cold", action="store_true", help="N = 1: add the cold-input regime of THIS workload (token workloads) as `cold`")
    ap.add_argument("--shard-of", type=int, default=0, metavar="N",
                    help="N = 1 GPU: additionally measure rank 0's share of this workload's batch split over N ranks (a 1/N-size launch), as a "
                         "tensor of its own and written into a whole-batch root tensor, as `shard` (the per-GPU term of the strong-scaling curve)")
    ap.add_argument("--shard-rank", type=int, default=0, help="which rank's share --shard-of measures")
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
    ap.add_argument("--devices", type=int, default=2, metavar="N",
                    help="N = 1 process: the e2e figure `e2e.list_to_devices` shards the list over N devices from this ONE process "
                         "(sharding.encode_on_devices); with fewer GPUs visible the entries repeat")
    ap.add_argument("--full-line", action="store_true",
                    help="print the FULL result object on stdout (the lab scripts and gpu_evidence.sh read it) instead of the compact "
                         "<= 4-KB line the driver parses; bench_full.json is written either way")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # Plain `python bench.py --gpus N`: become the launcher.  Nothing in this process has touched HIP or imported
        # torch yet, and the ranks are CHILD processes (never an exec): torch.distributed.run starts one rank per GPU,
        # rank 0 prints the one JSON line, which is relayed unchanged; exit code = the launcher's.
        sys.exit(self_launch(args.gpus))
    requested_gpus = args.gpus
    if world != args.gpus:
        if args.gather > 0 and args.gpus > 1:
            # a gather figure is only worth reporting for the world it was asked for: a launcher that started
            # another number of ranks than --gpus must not produce a line that reads as the N-GPU gather
            sys.exit("bench.py: --gpus %d --gather %d, but the launcher started %d ranks (WORLD_SIZE): refusing to report a gather for another world size"
                     % (args.gpus, args.gather, world))
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
    backend_note = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            # RCCL carries the barrier and the tiny timing reductions only -- the path shards by sequence, there is no data-path collective -- so
            # a box whose RCCL does not come up must not cost the scaling curve: the same run then goes on over gloo, and the line says so
            try:
                dist.init_process_group("nccl", device_id=dev)  # RCCL
                probe = torch.ones(1, device=dev)
                dist.all_reduce(probe)
                torch.cuda.synchronize()
                if int(probe.item()) != world:
                    raise RuntimeError("all_reduce of ones over %d ranks returned %r" % (world, probe.item()))
            except Exception as ex:  # noqa: BLE001 (whatever RCCL / the HIP runtime raise)
                if args.gather:
                    raise  # (the gather figures ARE RCCL traffic: nothing to report without it)
                backend_note = "nccl failed (%s: %s); barrier + timing reductions over gloo" % (type(ex).__name__, str(ex).splitlines()[0][:120] if str(ex) else "")
                print("bench.py rank %d: %s" % (rank, backend_note), file=sys.stderr, flush=True)
                try:
                    if dist.is_initialized():
                        dist.destroy_process_group()
                except Exception:  # noqa: BLE001
                    pass
                backend = "gloo"
                dist.init_process_group("gloo")
        else:
            dist.init_process_group(backend)
    red_dev = dev if backend == "nccl" else torch.device("cpu")  # where the tiny timing reductions live

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

    lib = capi.load()
    stream = torch.cuda.current_stream()
    b = Batch(args.workload, lib, dev, stream, n=n, first=first)
    chars, offsets, total = b.chars, b.offsets, b.total
    desc, C, dt_code, sz, tdt = b.desc, b.C, b.dt_code, b.sz, b.tdt
    d_chars, d_offs, out, out_bytes, algo_bytes, sh = b.d_chars, b.d_offs, b.out, b.out_bytes, b.algo_bytes, b.sh
    step = b.step

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed: validate lengths, run once, check a size-independent property and (full-size single-rank batches) the reference's folds
    check = b.check()

    # write-bandwidth yardstick: a plain fill (one 1-KiB store per wave, one aligned 4-KiB chunk per workgroup, blocks in
    # address order) over the same output buffer, in its BEST-KNOWN configuration: 3 resident workgroups per CU (unused
    # LDS as the cap; profiles/r01/fill_occupancy.txt: 6.84 TB/s uncapped, 7.38 at 3 per CU), and uncapped for reference.
    # Measured BEFORE the warm-up steps: its launches also lift the clocks out of idle.
    capi.check(lib.bsq_tuning_set(b"fill_mode", 1))
    fill_bytes = (out_bytes // 16) * 16
    fill_gbps = {}
    for name, pad in (("uncapped", 0), ("3_workgroups_per_cu", 53000)):
        capi.check(lib.bsq_tuning_set(b"fill_pad", pad))
        fill_gbps[name] = fill_bytes / (timed_loop(lambda: capi.check(lib.bsq_fill_device(out.data_ptr(), fill_bytes, 0, sh)), 5, 10, stream) * 1e-3) / 1e9
    capi.check(lib.bsq_tuning_set(b"fill_pad", 0))
    fill_best = max(fill_gbps.values())
    # read + write yardstick of the token workloads: the kernel's stream shape (one wave = one aligned 4-KiB chunk of the
    # output + its share of the characters, two dependent load steps like offsets -> characters) with none of its work
    mix_gbps = None
    if op in ("tokenize", "augment+tokenize") and batch_first and sz == 1 and out_bytes % 4096 == 0 and total >= 16:
        src_bytes = (total // 16) * 16
        mix_ms = timed_loop(lambda: capi.check(lib.bsq_copy_mix_device(out.data_ptr(), out_bytes, d_chars.data_ptr(), src_bytes, 1, 1, sh)), 20, 10, stream)
        mix_gbps = (src_bytes + out_bytes) / (mix_ms * 1e-3) / 1e9
    # what one event pair around ONE tiny launch reads: the floor under every per-step event time below
    tiny = torch.empty(4096, dtype=torch.uint8, device=dev)
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for ea, eb in pairs:
        ea.record(stream)
        capi.check(lib.bsq_fill_device(tiny.data_ptr(), 4096, 0, sh))
        eb.record(stream)
    torch.cuda.synchronize()
    event_floor_ms = float(np.median([ea.elapsed_time(eb) for ea, eb in pairs]))

    ramp(step, stream)
    for _ in range(args.warmup):
        step()
    # THE timed region: exactly K steps between barriers.  One pair of HIP events around the same K launches (recorded on
    # the launch stream) gives the kernel time of a step for the roofline -- rocprofv3's average kernel durations OF THESE K STEPS
    # (`timed.first_step_index`: scripts/make_traffic.py keeps exactly those dispatches of the trace) add up to it.
    la, lb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    first_timed_step = b.step_calls
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
    for ea, eb in ev:
        ea.record(stream)
        step()
        eb.record(stream)
    torch.cuda.synchronize()
    kern_ms = [ea.elapsed_time(eb) for ea, eb in ev]
    kern_avg_ms = float(np.mean(kern_ms))
    if os.environ.get("BSQ_BENCH_DUMP"):  # per-step device times, for variance hunting
        print("per-step ms:", " ".join("%.3f" % v for v in kern_ms), file=sys.stderr)
    if op == "augment+tokenize":
        d_chars.copy_(torch.from_numpy(chars).to(dev))

    # Sustained: the same step looped for at least one second of wall clock (clocks, thermals), untimed for `value`.
    sustained = None
    if world == 1 and not args.no_sustained:
        sustained = sustained_loop(b, 1.0, args.steps, loop_ms, stream)
    step_calls_main = b.step_calls
    cold = None
    if world == 1 and args.cold and (op in ("tokenize", "augment+tokenize") or args.workload in COLD_ONEHOT):
        cold = cold_regime(b, args.steps, 1.0, stream)

    gather_info = None
    if world > 1 and args.gather > 0:
        from bioseq_amd import sharding
        seq_first = op == "onehot" or (op == "tokenize" and not batch_first)
        axis = 1 if seq_first else 0
        out_c = out.contiguous() if backend == "nccl" else out.cpu().contiguous()  # gloo smoke runs move host tensors

        # what every assembled batch must be: THIS rank's own encode of the whole job batch (sequences [0, n_job) of the stream) in
        # one single-rank pass -- compared by content (the three 64-bit folds), not by shape
        whole = Batch(args.workload, lib, dev, stream, n=n_job, first=0)
        whole.step()
        torch.cuda.synchronize()
        want = fold_device(whole.out)
        want_shape = tuple(whole.out.shape)
        if whole.full_size and golden_folds().get(args.workload):
            g = golden_folds()[args.workload]
            assert want == (g["xor"], g["sum"], g["wsum"]), "single-rank encode of the job batch differs from the reference's folds"
        del whole
        torch.cuda.empty_cache()

        def timed(fn, check_axis=None):
            ms = []
            for _ in range(args.gather + 1):
                barrier()
                g0 = time.perf_counter()
                full = fn()
                barrier()
                ms.append((time.perf_counter() - g0) * 1e3)
                if full is not None:  # (None: this rank is not the root of a rooted form)
                    assert tuple(full.shape) == want_shape, (tuple(full.shape), want_shape)
                    assert fold_device(full) == want, "an assembled batch differs from the single-rank encode of the whole batch"
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
        try:
            nccl_version = ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None
        except Exception:
            nccl_version = None
        if dist.get_world_size() != requested_gpus:
            raise SystemExit("bench.py: the process group has %d ranks, --gpus asked for %d: no gather line" % (dist.get_world_size(), requested_gpus))
        gather_info = {"rccl_world_size": dist.get_world_size(), "backend": backend, "rccl_version": nccl_version,
                       "shard_bytes_this_rank": out_bytes, "staging_cap_bytes_direct": 256 << 20,
                       "bytes_received_per_rank": recv_bytes, "forms": forms, "note": "encode time excluded; mean of %d "
                       "assemblies after one warm-up, MAX over ranks" % args.gather,
                       "check": "every assembled batch (every form, every repetition, on every rank that receives it) has the three 64-bit folds "
                                "of the single-rank encode of the whole job batch"}

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
        traffic = traffic_of(args.workload)
        kernel_name = b.kernel_name()
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
            "config": {"workload": b.describe(),
                "sequences_per_gpu": n, "padlen": P, "channels": C, "input_chars_per_gpu": total,
                "output_bytes_per_gpu": out_bytes,
                "sharding": ("by sequence, no collective; weak scaling: every rank encodes its own %d-sequence batch" % n)
                            if args.scaling == "weak" else
                            ("by sequence, no collective; strong scaling: ONE %d-sequence batch split over %d ranks "
                             "(sharding.shard_bounds; rank 0 holds %d)" % (n_job, world, n)),
                "job_sequences": n_job,
                "rccl_world_size": dist.get_world_size() if world > 1 else 1,
                "backend": (backend + (" (RCCL)" if backend == "nccl" else "") + (": " + backend_note if backend_note else "")) if world > 1 else "none (single process)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS,
                         "frac_wall": algo_bytes / (wall_max / args.steps) / 1e9 / HBM_PEAK_GBPS,  # from ms_per_step (host clock)
                         "traffic": traffic, "traffic_stale": traffic_stale(args.workload, lib), "kernel": kernel_name,
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
        res["check"] = check
        res["build_id"] = lib.bsq_build_id().decode()  # of the loaded libbsq_hip.so (profiles/traffic.json records the one it was measured on)
        # which step() calls of this process were the K timed ones (the rocprofv3 summaries under profiles/ keep exactly those dispatches)
        res["timed"] = {"first_step_index": first_timed_step, "steps": args.steps, "step_calls_total": step_calls_main}
        if sustained is not None:
            res["sustained"] = sustained
        if cold is not None:
            res["cold"] = cold
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
            res["e2e"] = e2e_python_surface(cfg, op, destchar, batch_first, chars, offsets, dev, args.devices)
        if world == 1 and not args.no_e2e and args.workload == "cfg3":
            try:
                res["loader"] = loader_epochs(dev)
            except Exception as ex:  # noqa: BLE001 -- a side figure must never cost the driver its line
                res["loader"] = {"error": repr(ex)}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(cfg, op, destchar, batch_first, chars, offsets, args.cpu_threads or None)
        # every other BASELINE workload in the same line (N = 1; default: only beside the headline workload)
        which = "none" if args.no_configs else (args.configs if args.configs is not None else ("all" if args.workload == "cfg3" else "none"))
        names = DEFAULT_CONFIGS if which == "all" else [] if which == "none" else [w for w in which.split(",") if w]
        if world == 1 and (names or args.shard_of > 1):
            del b, d_chars, d_offs, out, step
            torch.cuda.empty_cache()
        if world == 1 and args.shard_of > 1:
            res["shard"] = run_shard(args.workload, args.shard_of, lib, dev, stream, args.steps, args.warmup, full_ms=loop_ms,
                                     full_cold_ms=(cold or {}).get("ms_per_step"), rank=args.shard_rank)
        if world == 1 and names:
            res["configs"] = {}
            for wname in names:
                if wname not in WORKLOADS:
                    raise SystemExit("unknown workload in --configs: %r" % wname)
                res["configs"][wname] = run_config(wname, lib, dev, stream, args.steps, args.warmup)
            if which == "all":
                # the per-GPU term of the 8-rank strong-scaling curve, on this one GPU (no multi-GPU node has been offered to the driver)
                for wname in SHARD_CONFIGS:
                    full = res["configs"].get(wname)
                    full_ms = loop_ms if wname == args.workload else (full or {}).get("ms_per_step_cache_resident", (full or {}).get("ms_per_step"))
                    full_cold = ((full or {}).get("cold") or {}).get("ms_per_step")
                    res["configs"]["%s_shard8" % wname] = run_shard(wname, 8, lib, dev, stream, args.steps, args.warmup, full_ms=full_ms, full_cold_ms=full_cold)
        emit(res, args.full_line)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
