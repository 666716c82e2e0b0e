"""Build container only (needs oracle/_ref): the product's HOST paths that need no device -- single-sequence `onehot_encode` (tokenize.h:188-216) and
`decode_tokens` (tokenize.h:131-183) -- against the reference's own C++ on random inputs.   python tests/stress/cpu_stress_single_decode.py 3000 5"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # tests/stress/ -> repo root
sys.path.insert(0, ROOT)
import numpy as np
import bioseq_amd as bsq
from oracle import oracle as O
ref = O.load_reference(); assert ref is not None
rng = np.random.default_rng(int(sys.argv[2]))
KEYS = [k for k in O.keys() if k != "BYTES"]
LET = b"ACGTNacgtnRYKMDEFHILPQSVWXBZUO"
n_one = n_dec = n_raise = 0
MAPPED = {}


def both(f, g):
    """results of the two calls, or the exceptions they raise (type + text) -- to be compared"""
    out = []
    for h in (f, g):
        try:
            out.append(("ok", h()))
        except Exception as ex:  # noqa: BLE001
            out.append(("raised", type(ex).__name__, str(ex)))
    return out


for trial in range(int(sys.argv[1])):
    key = KEYS[rng.integers(len(KEYS))]
    eos, bos, pad = (int(x) for x in rng.integers(0, 2, 3))
    r, t = ref.Tokenizer(key, eos, bos, pad), bsq.Tokenizer(key, eos, bos, pad)
    # single sequence: only MAPPED letters (an unmapped byte makes the reference write offp[-1]: SURVEY 8 f-4)
    if key not in MAPPED:  # letters the reference itself maps under this key (one-hot of the single letter has a one)
        plain = ref.Tokenizer(key, 0, 0, 0)
        MAPPED[key] = bytes(c for c in LET if plain.batch_onehot_encode([bytes([c])], padlen=1, destchar="b").any())
    mapped = MAPPED[key]
    L = int(rng.integers(0, 40))
    seq = bytes(rng.choice(list(mapped), size=L).astype(np.uint8)) if mapped and L else b""
    padlen = int(rng.integers(0, 50))
    for d in "BHIFD":
        for form in (seq, seq.decode(), bytearray(seq)):
            a, b = both(lambda: r.onehot_encode(form, padlen, d), lambda: t.onehot_encode(form, padlen, d))
            if a[0] == "raised" or b[0] == "raised":
                assert a == b, (key, eos, bos, pad, seq, padlen, d, type(form), a, b)
                n_raise += 1
                continue
            a, b = a[1], b[1]
            assert a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes(), (key, eos, bos, pad, seq, padlen, d, type(form))
            n_one += 1
    # decode: token matrices of valid ids (incl. BOS / EOS / PAD where they exist), 1-D and 2-D, several integer types
    C = t.alphabet_size()
    toks = rng.integers(0, C, size=(int(rng.integers(1, 6)), int(rng.integers(1, 30))))
    for dt in (np.int8, np.int16, np.int32, np.int64, np.uint8):
        if C > 127 and dt == np.int8:
            continue
        x = toks.astype(dt)
        assert r.decode_tokens(x) == t.decode_tokens(x), (key, eos, bos, pad, x)
        assert r.decode_tokens(x[0]) == t.decode_tokens(x[0])
        n_dec += 2
print("single-sequence one-hot and decode vs the compiled reference: %d + %d comparisons (+ %d calls that raise the same error in both), all equal" % (n_one, n_dec, n_raise))
