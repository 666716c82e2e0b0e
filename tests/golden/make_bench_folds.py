#!/usr/bin/env python3
"""Golden 64-bit folds of the FULL-SIZE outputs of every bench.py workload, from the REAL reference (build container only:
needs /root/reference compiled into oracle/_ref by `make -C oracle ref`; ~8 GB of RAM, a few minutes).

bench.py checks every workload it times against these -- untimed, on the device, and never against the product's own kernels:

    fold(bytes) = (xor, sum, wsum) over the little-endian uint64 words w_i of the array's bytes in C order (zero-padded to 8):
                  xor = XOR_i w_i,   sum = SUM_i w_i mod 2^64,   wsum = SUM_i w_i * (2 i + 1) mod 2^64   (position-dependent)

cfg5aug: the batch is first mutated by the numpy twin of the augmentation stream (tests/test_augment.py: twin -- seed 1, chain 1,
frac 0.5, exactly what bench.py's first step does), the fold of the mutated characters is stored too, and the tokens are the
reference's batch_tokenize of the mutated sequences.

`<w>_shard8` (round 5): rank 0's share (sharding.shard_bounds: the first n/8 sequences) of a STRONG-scaling split over 8 ranks of cfg3,
cfg4f, cfg4b and cfg5aug -- the per-GPU term of the 1/2/4/8 curve, which bench.py measures on one GPU -- encoded by the reference as
a batch of its own (`--shards-only` adds just these to the existing file).

Writes tests/golden/bench_folds.json (data only).
"""
import importlib.util
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
import cbioseq  # noqa: E402  (the compiled reference)


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


synth = _load("bsq_synth", os.path.join(ROOT, "bioseq_amd", "synth.py"))
M64 = (1 << 64) - 1


def fold(arr):
    b = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
    if b.size % 8:
        b = np.concatenate([b, np.zeros(8 - b.size % 8, dtype=np.uint8)])
    w = b.view("<u8")
    x = int(np.bitwise_xor.reduce(w)) if w.size else 0
    s = ws = 0
    step = 1 << 24
    with np.errstate(over="ignore"):
        for i in range(0, w.size, step):
            c = w[i:i + step]
            s = (s + int(c.sum(dtype=np.uint64))) & M64
            k = np.arange(i, i + c.size, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
            ws = (ws + int((c * k).sum(dtype=np.uint64))) & M64
    return {"xor": "%016x" % x, "sum": "%016x" % s, "wsum": "%016x" % ws, "nbytes": int(arr.nbytes)}


def twin_cfg5(chars, offs, seed, chain_len=1, frac=0.5):
    """tests/test_augment.py: twin, vectorised over nothing -- 262 144 short Python iterations."""
    ta = _load("bsq_test_augment_twin", os.path.join(ROOT, "tests", "test_augment.py"))
    normrows = np.load(os.path.join(HERE, "blosum_normrows.npy"))
    return ta.twin(chars, offs, chain_len, frac, seed, normrows)


SHARD_WORKLOADS = {"cfg3": ("cfg3", "onehot", "f"), "cfg4f": ("cfg4", "onehot", "f"), "cfg4b": ("cfg4", "onehot", "B"),
                   "cfg5aug": ("cfg5", "augment+tokenize", "B")}


def shards(out, world=8):
    for w, (cname, op, destchar) in SHARD_WORKLOADS.items():
        c = synth.CONFIGS[cname]
        base, extra = divmod(c["n"], world)
        n = base + (1 if extra else 0)  # sharding.shard_bounds(n, world, 0)
        chars, offs = synth.synth_packed(c["seed"], n, c["lo"], c["hi"], c["letters"])
        tok = cbioseq.Tokenizer(c["key"], c["eos"], c["bos"], c["padchar"])
        name = "%s_shard%d" % (w, world)
        if op == "onehot":
            out[name] = fold(tok.batch_onehot_encode(synth.unpack(chars, offs), padlen=c["padlen"], destchar=destchar, nthreads=8))
        else:
            mut = twin_cfg5(chars, offs, seed=1)
            out[name] = fold(tok.batch_tokenize(synth.unpack(mut, offs), padlen=c["padlen"], batch_first=True, nthreads=8))
            out[name]["mutated_chars"] = fold(mut)
            out[name]["mutated_sequences"] = int((np.add.reduceat((mut != chars).astype(np.int64), offs[:-1]) > 0).sum())
        out[name]["sequences"] = n
        out[name]["what"] = "rank 0's shard (the first %d sequences) of %s split over %d ranks, encoded by the reference as its own batch" % (n, w, world)
        print(name, "done", flush=True)


def main():
    if "--shards-only" in sys.argv:
        path = os.path.join(HERE, "bench_folds.json")
        out = json.load(open(path))
        shards(out)
        with open(path, "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)
        print("written", path)
        return
    out = {"definition": "fold = (xor, sum mod 2^64, sum of w_i * (2 i + 1) mod 2^64) over the little-endian uint64 words of the "
                         "array's bytes in C order, zero-padded to a multiple of 8; hex"}

    def batch(name):
        c = synth.CONFIGS[name]
        chars, offs = synth.synth_packed(c["seed"], c["n"], c["lo"], c["hi"], c["letters"])
        return c, chars, offs, cbioseq.Tokenizer(c["key"], c["eos"], c["bos"], c["padchar"])

    c, chars, offs, tok = batch("cfg2")
    seqs = synth.unpack(chars, offs)
    out["cfg2"] = fold(tok.batch_tokenize(seqs, padlen=c["padlen"], batch_first=True, nthreads=8))
    out["cfg2sf"] = fold(tok.batch_tokenize(seqs, padlen=c["padlen"], batch_first=False, nthreads=8))      # the reference's default layout
    out["cfg3b"] = fold(tok.batch_onehot_encode(seqs, padlen=c["padlen"], nthreads=8))                     # the reference's default dtype 'B'
    oh = tok.batch_onehot_encode(seqs, padlen=c["padlen"], destchar="f", nthreads=8)
    out["cfg3"] = fold(oh)
    out["cfg3bcl"] = fold(np.ascontiguousarray(oh.transpose(1, 2, 0)))                                     # (B, C, P): einops 'l b c -> b c l' (loaders.py:74)
    del oh, seqs
    print("cfg2/3 done", flush=True)
    c, chars, offs, tok = batch("cfg4")
    seqs = synth.unpack(chars, offs)
    out["cfg4f"] = fold(tok.batch_onehot_encode(seqs, padlen=c["padlen"], destchar="f", nthreads=8))
    out["cfg4b"] = fold(tok.batch_onehot_encode(seqs, padlen=c["padlen"], destchar="B", nthreads=8))
    del seqs
    print("cfg4 done", flush=True)
    c, chars, offs, tok = batch("cfg5")
    out["cfg5"] = fold(tok.batch_tokenize(synth.unpack(chars, offs), padlen=c["padlen"], batch_first=True, nthreads=8))
    mut = twin_cfg5(chars, offs, seed=1)
    out["cfg5aug"] = fold(tok.batch_tokenize(synth.unpack(mut, offs), padlen=c["padlen"], batch_first=True, nthreads=8))
    out["cfg5aug"]["mutated_chars"] = fold(mut)
    out["cfg5aug"]["mutated_sequences"] = int((np.add.reduceat((mut != chars).astype(np.int64), offs[:-1]) > 0).sum())
    out["cfg5aug"]["what"] = "seed 1, chain_len 1, augment_frac 0.5: numpy twin of the augmentation stream, then the reference's batch_tokenize"
    c, chars, offs, tok = batch("cfg1")
    out["cfg1oh"] = fold(tok.batch_onehot_encode(synth.unpack(chars, offs), padlen=c["padlen"], destchar="f"))
    shards(out)
    with open(os.path.join(HERE, "bench_folds.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("written", os.path.join(HERE, "bench_folds.json"))


if __name__ == "__main__":
    main()
