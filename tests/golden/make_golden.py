#!/usr/bin/env python3
"""Generate the committed golden fixtures from the REAL reference (build container only).

Needs /root/reference and oracle/_ref/cbioseq*.so (`make -C oracle ref`): the reference's own
src/tokenize.cpp + src/omp.cpp + src/fxstats.cpp compiled in place, plus the reference's pure-Python package
imported from /root/reference for the module-level facade (bioseq/__init__.py:36-168) and the
BLOSUM table (bioseq/blosum.py:36-48).  Only DATA is written here (inputs are regenerated from
bioseq_amd/synth.py seeds; expected outputs are stored as raw arrays or sha256 digests).

    python tests/golden/make_golden.py            # small fixtures (seconds)
    python tests/golden/make_golden.py --full     # + cfg2/3/4/5 full-size digests (minutes, ~12 GB RAM)

Files written next to this script:
    alphabets.json   LUTs, ids and decode tables of every key x (eos,bos,padchar)
    kats.json        known-answer digests: README vector, cfg1..cfg5, dirty table
    small_cases.npz  raw expected arrays for small ragged/dirty batches (all dtypes, layouts, masks)
    facade.npz/json  outputs of bioseq.onehot_encode / f_encode and the tokenizer-dict key lists
    blosum_normrows.npy
    flatfile_small.fa/.ff  a small FASTA/FASTQ text and the FlatFile the reference writes from it
    augment_law.json  (position x new residue) counts of the reference's augment_seq on a mixed sequence
    decode.json.gz, decode_tokens.npz, single.json.gz   row f-4: decode_tokens strings and single-sequence one-hot digests
"""
import argparse
import gzip
import hashlib
import itertools
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
sys.path.insert(1, "/root/reference")
sys.path.insert(2, ROOT)

import cbioseq  # noqa: E402  (the compiled reference)
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("bsq_synth", os.path.join(ROOT, "bioseq_amd", "synth.py"))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)

KEYS = ["AMINO", "AMINO20", "BYTES", "C", "DAYHOFF", "DNA", "DNA4", "DNA5", "DNAMETH", "KETO", "LIA10",
        "LIB10", "MURPHY", "PROTEIN", "PURPYR", "SEB10", "SEB14", "SEB6", "SEB8", "SEV10"]
COMBOS = list(itertools.product([0, 1], repeat=3))  # (eos, bos, padchar) = positional ctor order


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def l1(b):
    return bytes(b).decode("latin-1")


def alphabets():
    out = {"keys": KEYS, "luts": {}, "meta": {}}
    for key in KEYS:
        plain = cbioseq.Tokenizer(key)
        lut = [-1] * 256
        for tok, members in plain.token_decoder().items():
            for byte in members:
                lut[byte] = tok
        out["luts"][key] = lut
        out["meta"][key] = {}
        for eos, bos, pad in COMBOS:
            t = cbioseq.Tokenizer(key.lower(), eos, bos, pad)
            m = dict(alphabet_size=t.alphabet_size(), bos=t.bos(), eos=t.eos(), pad=t.pad(), nchars=t.nchars(),
                     key=t.key, is_padded=t.is_padded(), includes_bos=t.includes_bos(),
                     includes_eos=t.includes_eos())
            if key != "BYTES":  # BYTES lookup strings are not valid UTF-8 -> lut()/token_map() raise in the reference
                m["token_map"] = t.token_map()
                m["lut"] = {str(k): v for k, v in t.lut().items()}
            if (eos, bos, pad) in ((0, 0, 0), (1, 1, 1)):
                m["decoder"] = {str(k): l1(v) for k, v in t.token_decoder().items()}
            out["meta"][key][f"{eos}{bos}{pad}"] = m
    with open(os.path.join(HERE, "alphabets.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"), sort_keys=True)


def kat_entry(arr):
    return dict(dtype=str(arr.dtype), shape=list(arr.shape), sum=float(arr.sum(dtype=np.float64)), sha256=sha(arr))


def kats(full):
    K = {}
    t = cbioseq.Tokenizer("DNA", True, True, True)
    r = t.batch_tokenize(["ACGT", "GGGG"], padlen=7, batch_first=True)
    K["readme"] = dict(tokens=r.tolist(), dtype=str(r.dtype), decoded=t.decode_tokens(r))

    def batch(name):
        c = synth.CONFIGS[name]
        chars, offs = synth.synth_packed(c["seed"], c["n"], c["lo"], c["hi"], c["letters"])
        meta = dict(total=int(offs[-1]), in_sha=sha(chars)[:16], lens_sha=sha(np.diff(offs).astype("<i8"))[:16])
        return c, synth.unpack(chars, offs), meta

    c, seqs, meta = batch("cfg1")
    tok = cbioseq.Tokenizer(c["key"], c["eos"], c["bos"], c["padchar"])
    K["cfg1"] = dict(input=meta, **kat_entry(tok.batch_tokenize(seqs, padlen=c["padlen"], batch_first=True, nthreads=1)))
    K["cfg1b"] = dict(input=meta, **kat_entry(tok.batch_tokenize(seqs, padlen=c["padlen"], batch_first=False)))

    # dirty table: 4 keys x 4 combos, tokens (batch_first int8) and one-hot (f32)
    c, seqs, meta = batch("dirty")
    K["dirty"] = dict(input=meta, rows={})
    for key in ("AMINO20", "SEB8", "DNA5", "DAYHOFF"):
        for eos, bos, pad in ((0, 0, 0), (1, 1, 1), (1, 0, 0), (0, 1, 1)):
            tok = cbioseq.Tokenizer(key, eos, bos, pad)
            K["dirty"]["rows"][f"{key}:{eos}{bos}{pad}"] = dict(
                tok=kat_entry(tok.batch_tokenize(seqs, padlen=c["padlen"], batch_first=True)),
                onehot=kat_entry(tok.batch_onehot_encode(seqs, padlen=c["padlen"], destchar="f")))
    if full:
        c, seqs, meta = batch("cfg2")
        tok = cbioseq.Tokenizer(c["key"], c["eos"], c["bos"], c["padchar"])
        K["cfg2"] = dict(input=meta, **kat_entry(tok.batch_tokenize(seqs, padlen=c["padlen"], batch_first=True, nthreads=8)))
        K["cfg2b"] = dict(input=meta, **kat_entry(tok.batch_tokenize(seqs, padlen=c["padlen"], batch_first=False, nthreads=8)))
        K["cfg3"] = dict(input=meta, **kat_entry(tok.batch_onehot_encode(seqs, padlen=c["padlen"], destchar="f", nthreads=8)))
        del seqs
        c, seqs, meta = batch("cfg5")
        tok = cbioseq.Tokenizer(c["key"], c["eos"], c["bos"], c["padchar"])
        K["cfg5"] = dict(input=meta, **kat_entry(tok.batch_tokenize(seqs, padlen=c["padlen"], batch_first=True, nthreads=8)))
        del seqs
        c, seqs, meta = batch("cfg4")
        tok = cbioseq.Tokenizer(c["key"], c["eos"], c["bos"], c["padchar"])
        K["cfg4-f"] = dict(input=meta, **kat_entry(tok.batch_onehot_encode(seqs, padlen=c["padlen"], destchar="f", nthreads=8)))
        K["cfg4-B"] = dict(input=meta, **kat_entry(tok.batch_onehot_encode(seqs, padlen=c["padlen"], destchar="B", nthreads=8)))
    else:  # keep previously generated full-size entries
        p = os.path.join(HERE, "kats.json")
        if os.path.exists(p):
            old = json.load(open(p))
            for k in ("cfg2", "cfg2b", "cfg3", "cfg5", "cfg4-f", "cfg4-B"):
                if k in old:
                    K[k] = old[k]
    with open(os.path.join(HERE, "kats.json"), "w") as f:
        json.dump(K, f, indent=1, sort_keys=True)


def small_cases():
    """Raw arrays.  Inputs: synth(seed, n, lo, hi, letters) -- regenerated by the tests."""
    arrays, index = {}, []
    rng = np.random.default_rng(12345)

    def add(name, arr, **desc):
        arrays[name] = arr
        index.append(dict(name=name, **desc))

    # A: dirty ragged protein batch incl. empty sequences; exact-fit (L+bos+eos == P) for the longest
    specA = dict(seed=707, n=37, lo=0, hi=40, letters=synth.DIRTY)
    chars, offs = synth.synth_packed(**specA)
    seqsA = synth.unpack(chars, offs)
    maxL = int(np.diff(offs).max())
    maskA = [(rng.random(len(s)) < 0.6).astype(np.uint8) if i % 4 else None for i, s in enumerate(seqsA)]
    arrays["A_mask_bytes"] = np.concatenate([m if m is not None else np.ones(len(s), np.uint8) for m, s in zip(maskA, seqsA)])
    n = 0
    for key in ("AMINO20", "SEB8", "DNA5", "KETO", "BYTES"):
        for eos, bos, pad in COMBOS:
            tok = cbioseq.Tokenizer(key, eos, bos, pad)
            P = maxL + eos + bos  # exact fit for the longest sequence
            for d in ("b", "h", "i", "q", "f", "d"):
                if n % 3 == 0 or (d in "bf"):  # thin the cross product, keep every int8/f32 case
                    for bf in (False, True):
                        add(f"A{n}", tok.batch_tokenize(seqsA, padlen=P, destchar=d, batch_first=bf),
                            batch=specA, op="tokenize", key=key, eos=eos, bos=bos, padchar=pad, padlen=P,
                            destchar=d, batch_first=bf)
                        n += 1
                    add(f"A{n}", tok.batch_onehot_encode(seqsA, padlen=P + 3, destchar=d),
                        batch=specA, op="onehot", key=key, eos=eos, bos=bos, padchar=pad, padlen=P + 3, destchar=d)
                    n += 1
                    if key != "BYTES":
                        add(f"A{n}", tok.batch_onehot_encode(seqsA, padlen=P, destchar=d, mask=maskA),
                            batch=specA, op="onehot", key=key, eos=eos, bos=bos, padchar=pad, padlen=P, destchar=d,
                            mask="A_mask_bytes")
                        n += 1
    # B: more sequences than one tile in both directions (B=150 > 64/128, P=200 > 64) -- tile-edge coverage
    specB = dict(seed=808, n=150, lo=0, hi=190, letters="ACGTNacgtnU")
    chars, offs = synth.synth_packed(**specB)
    seqsB = synth.unpack(chars, offs)
    for key, (eos, bos, pad), d in (("DNA", (1, 1, 1), "b"), ("DNA5", (0, 1, 0), "f"), ("DNA", (1, 0, 1), "h")):
        tok = cbioseq.Tokenizer(key, eos, bos, pad)
        for bf in (False, True):
            add(f"B{n}", tok.batch_tokenize(seqsB, padlen=193, destchar=d, batch_first=bf), batch=specB, op="tokenize",
                key=key, eos=eos, bos=bos, padchar=pad, padlen=193, destchar=d, batch_first=bf)
            n += 1
        add(f"B{n}", tok.batch_onehot_encode(seqsB, padlen=193, destchar=d), batch=specB, op="onehot", key=key,
            eos=eos, bos=bos, padchar=pad, padlen=193, destchar=d)
        n += 1
    np.savez_compressed(os.path.join(HERE, "small_cases.npz"), **arrays)
    with open(os.path.join(HERE, "small_cases.json"), "w") as f:
        json.dump(index, f, separators=(",", ":"))


def facade():
    import bioseq  # the reference's pure-Python package, /root/reference/bioseq/__init__.py
    J = dict(bkeys=list(bioseq.bkeys), default_keys=sorted(bioseq.default_tokenizers),
             dict_sizes={n: len(getattr(bioseq, n)) for n in
                         ("default_tokenizers", "pbeos_tokenizers", "beos_tokenizers", "pbos_tokenizers",
                          "bos_tokenizers", "peos_tokenizers", "eos_tokenizers", "pos_tokenizers",
                          "total_tokenizer_dict")},
             get_tokenizer_dict={f"{b}{e}{p}": [bioseq.get_tokenizer_dict(b, e, p)["DNA"].includes_bos(),
                                                bioseq.get_tokenizer_dict(b, e, p)["DNA"].includes_eos(),
                                                bioseq.get_tokenizer_dict(b, e, p)["DNA"].is_padded()]
                                 for b, e, p in COMBOS},
             named={n: [getattr(bioseq, n).key, getattr(bioseq, n).alphabet_size()] for n in
                    ("DNATokenizer", "AmineTokenizer", "Reduced6Tokenizer", "Reduced8Tokenizer",
                     "Reduced10Tokenizer", "Reduced14Tokenizer", "DayhoffTokenizer", "LIATokenizer", "LIBTokenizer")})
    A = {}
    seqs = ["ACGT", "GGN", "", "acgtACGT"]
    A["f_encode_dna_bos_p10"] = bioseq.f_encode(seqs, key="dna", bos=True, padlen=10)
    A["f_encode_prot_pbeos_f_bf"] = bioseq.f_encode(["MKV", "ACDEFGHIKL"], key="PROTEIN", bos=True, eos=True,
                                                   padchar=True, padlen=13, destchar="f", batch_first=True)
    t = bioseq.onehot_encode(bioseq.pbeos_tokenizers["DNA"], seqs, padlen=11, destchar="f", batch_first=True,
                             to_pytorch=True)
    A["onehot_encode_torch_bf"] = t.numpy().copy()
    J["onehot_encode_torch_bf"] = dict(dtype=str(t.dtype), contiguous=bool(t.is_contiguous()))
    A["onehot_encode_numpy_sf"] = bioseq.onehot_encode(bioseq.beos_tokenizers["AMINO20"], ["MKV", b"ACDEFGHIKL", bytearray(b"WY")],
                                                       padlen=12, destchar="h")
    A["onehot_encode_numpy_bf"] = bioseq.onehot_encode(bioseq.pos_tokenizers["SEB8"], ["MKV", "ACDEFGHIKL", ""], padlen=10,
                                                       destchar="d", batch_first=True)
    t = bioseq.f_encode(["ACGT", "NNNN", "acg"], key="DNA5", eos=True, padchar=True, padlen=6, destchar="i", to_pytorch=True)
    A["f_encode_torch_sf_i32"] = t.numpy().copy()
    J["f_encode_torch_sf_i32"] = dict(dtype=str(t.dtype), contiguous=bool(t.is_contiguous()))
    t = bioseq.f_encode("ACGTTGCA", key="DNA", bos=True, padlen=10, destchar="f", to_pytorch=True)
    A["f_encode_single_torch"] = t.numpy().copy()
    # single-sequence path: str -> 'B' is uint8 there, shape (max(L,padlen)+bos+eos, C)
    A["single_str_default"] = bioseq.f_encode("ACGT", key="DNA")
    A["single_pbeos_p8"] = bioseq.pbeos_tokenizers["DNA"].onehot_encode("ACGT", 8, "f")
    A["single_bytes_default"] = bioseq.DNATokenizer.onehot_encode(b"ACGTA")
    np.savez_compressed(os.path.join(HERE, "facade.npz"), **A)
    with open(os.path.join(HERE, "facade.json"), "w") as f:
        json.dump(J, f, indent=1, sort_keys=True)
    from bioseq import blosum
    np.save(os.path.join(HERE, "blosum_normrows.npy"), blosum.normrows.astype("<f8"))


DEC_KEYS = ["DNA", "AMINO20", "SEB8", "DNA5", "BYTES", "KETO", "DAYHOFF"]
SINGLE_KEYS = ["DNA", "AMINO20", "SEB8", "PURPYR"]


def decode_and_single():
    """Row f-4 pinned to the reference.

    decode.json.gz / decode_tokens.npz: seeded token matrices (stored) and what the REFERENCE's `decode_tokens`
    (src/tokenize.h:131-183) returns for them as 2-D arrays of every item size (1/2/4/8, signed and unsigned -- all must
    agree, asserted here), as 1-D rows, through a transposed view and through a column-strided view.
    single.json.gz: the reference's single-sequence `onehot_encode` (src/tokenize.h:188-216, src/tokenize.cpp:24-51) for
    str / bytes / bytearray x padlen {0, L, L+5} x the case-masked dtype characters (+ the per-type default), as
    dtype / shape / sha256 of the returned array.  Only letters the alphabet maps are used: an unmapped byte makes the
    reference write before the row (tokenize.h:203-206)."""
    rng = np.random.default_rng(20260301)
    arrays, dec = {}, []
    n = 0
    for key in DEC_KEYS:
        for eos, bos, pad in COMBOS:
            tok = cbioseq.Tokenizer(key, eos, bos, pad)
            ids = sorted(k for k in tok.token_decoder().keys() if 0 <= k < 128)
            ids += [i for i, on in ((tok.bos(), bos), (tok.eos(), eos), (tok.pad(), pad)) if on]
            for shape in ((1, 1), (3, 64), (5, 65), (7, 200), (70, 33)):
                toks = rng.choice(np.array(ids, dtype=np.int64), size=shape)
                want = tok.decode_tokens(toks)
                for dt in (np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.int64, np.uint64):
                    if toks.max() <= np.iinfo(dt).max:
                        assert tok.decode_tokens(toks.astype(dt)) == want, (key, dt)
                rows = [tok.decode_tokens(toks[r].astype(np.int32)) for r in range(shape[0])]
                assert rows == want
                name = "T%d" % n
                n += 1
                arrays[name] = toks.astype(np.int16 if toks.max() > 127 else np.int8)
                dec.append(dict(name=name, key=key, eos=eos, bos=bos, padchar=pad,
                                decoded=want,
                                transposed=tok.decode_tokens(toks.astype(np.int16).T),
                                strided=tok.decode_tokens(toks.astype(np.int32)[:, ::2])))
    np.savez_compressed(os.path.join(HERE, "decode_tokens.npz"), **arrays)
    with gzip.GzipFile(os.path.join(HERE, "decode.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(dec, separators=(",", ":")).encode())

    single = []
    for key in SINGLE_KEYS:
        plain = cbioseq.Tokenizer(key)
        letters = "".join(sorted(chr(b) for t, members in plain.token_decoder().items() if t >= 0 for b in members
                                 if chr(b).isalpha()))
        seqs = ["", letters[0], "".join(rng.choice(list(letters), size=37)), "".join(rng.choice(list(letters), size=130))]
        for eos, bos, pad in COMBOS:
            tok = cbioseq.Tokenizer(key, eos, bos, pad)
            for si, seq in enumerate(seqs):
                for padlen in (0, len(seq), len(seq) + 5):
                    for kind, obj in (("str", seq), ("bytes", seq.encode()), ("bytearray", bytearray(seq.encode()))):
                        dts = ["", "B", "H", "I", "F", "D", "b", "h", "i", "f", "d"] if kind == "str" else ["", "f", "H"]
                        for dt in dts:
                            a = tok.onehot_encode(obj, padlen, dt) if dt else tok.onehot_encode(obj, padlen)
                            single.append([key, eos, bos, pad, si, padlen, kind, dt, str(a.dtype), list(a.shape),
                                           int(a.sum()), sha(a)[:24]])
        single.append(dict(key=key, seqs=seqs))
    with gzip.GzipFile(os.path.join(HERE, "single.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(single, separators=(",", ":")).encode())


AUG_SEQ = "AWHKCLGPSTYV"
AUG_N = 200000


def augment_law():
    """augment_law.json: the REFERENCE's `augment_seq` (bioseq/blosum.py:63-87) run AUG_N times on one heterogeneous
    sequence, chain_len 1, as a (position x new residue) count table -- data only.  The mutated position is NOT uniform:
    the reject-until-changed loop makes it proportional to 1 - normrows[r, r] (W: almost never, A: often)."""
    import time
    from bioseq import blosum
    letters = "".join(blosum.aa_array)
    counts = np.zeros((len(AUG_SEQ), len(letters)), dtype=np.int64)
    t0 = time.time()
    for _ in range(AUG_N):
        out = blosum.augment_seq(AUG_SEQ, 1)
        (pos,) = [i for i, (a, b) in enumerate(zip(AUG_SEQ, out)) if a != b]  # exactly one residue changes
        counts[pos, letters.index(out[pos])] += 1
    with open(os.path.join(HERE, "augment_law.json"), "w") as f:
        json.dump(dict(seq=AUG_SEQ, n=AUG_N, chain_len=1, letters=letters, counts=counts.tolist(),
                       generator="bioseq.blosum.augment_seq, module rng default_rng(72) as imported", seconds=round(time.time() - t0)),
                  f, separators=(",", ":"))


def flatfile_fixture():
    """tests/golden/flatfile_small.{fa,ff}: the .ff is written by the reference's own FlatFile
    (src/fxstats.cpp:33-64) from the small FASTA/FASTQ text next to it."""
    fa = os.path.join(HERE, "flatfile_small.fa")
    with open(fa, "w") as f:
        f.write(">sp|P1|first protein\nMKVLAAGIVGLLLA\nQPSNA\n>empty\n\n>third x\nACDEFGHIKLMNPQRSTVWY\nacdef\n"
                "@read1\nACGTN\n+\nIIIII\n>last\nWWWW\n")
    cbioseq.FlatFile(fa, os.path.join(HERE, "flatfile_small.ff"))



if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--augment-law", action="store_true", help="re-run the reference's augment_seq 200 000 times (about a minute)")
    a = ap.parse_args()
    alphabets()
    small_cases()
    facade()
    flatfile_fixture()
    decode_and_single()
    kats(a.full)
    if a.augment_law or not os.path.exists(os.path.join(HERE, "augment_law.json")):
        augment_law()
    print("golden fixtures written to", HERE)

