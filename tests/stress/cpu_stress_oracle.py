"""Build container only (needs oracle/_ref): the C oracle against the reference's own C++ on random batches -- every key, flag combination, dtype,
layout, masks, dirty alphabets (never bytes >= 0x80 or BYTES as int8: SURVEY 8c).   python tests/stress/cpu_stress_oracle.py 1500 99"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # tests/stress/ -> repo root
sys.path.insert(0, ROOT)
import numpy as np
from bioseq_amd import synth
from oracle import oracle as O
O.build()
ref = O.load_reference(); assert ref is not None
rng = np.random.default_rng(int(sys.argv[2]))
KEYS = O.keys()
ALPH = [synth.DIRTY, synth.AA, "ACGT", "ACGTNacgtn", synth.DIRTY + "".join(chr(c) for c in range(33, 64))]
n_cmp = 0
for trial in range(int(sys.argv[1])):
    n, hi = int(rng.integers(0, 400)), int(rng.integers(0, 300))
    chars, offs = synth.synth_packed(int(rng.integers(1 << 30)), n, int(rng.integers(0, hi + 1)), hi, ALPH[rng.integers(len(ALPH))])
    seqs = synth.unpack(chars, offs)
    key = KEYS[rng.integers(len(KEYS))]
    eos, bos, pad = (int(x) for x in rng.integers(0, 2, 3))
    P = hi + eos + bos + int(rng.integers(0, 5))
    if P <= 0: continue
    mask = [(rng.random(len(s)) < 0.5).astype(np.uint8) if rng.random() < 0.7 else None for s in seqs] if rng.random() < 0.5 else None
    if mask is not None and all(m is None for m in mask): mask = None
    r, o = ref.Tokenizer(key, eos, bos, pad), O.OracleTokenizer(key, eos, bos, pad)
    for d in "bhiqfd":
        if key == "BYTES" and d == "b": continue
        for bf in (False, True):
            a = r.batch_tokenize(seqs, padlen=P, destchar=d, batch_first=bf); b = o.batch_tokenize(seqs, padlen=P, destchar=d, batch_first=bf)
            assert a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes(), (trial, key, eos, bos, pad, d, bf)
        a = r.batch_onehot_encode(seqs, padlen=P, destchar=d, mask=mask); b = o.batch_onehot_encode(seqs, padlen=P, destchar=d, mask=mask)
        assert a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes(), (trial, key, eos, bos, pad, d, "onehot")
        n_cmp += 3
print("oracle vs compiled reference: %d random batches, %d array comparisons, all bit-exact" % (int(sys.argv[1]), n_cmp))
