"""CPU, property-based (hypothesis): invariants of the encode semantics on the oracle, and -- in the
build container -- the oracle against the reference's own C++ on generated batches."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

ALPHA = "ACDEFGHIKLMNPQRSTVWYacgtnXBZUO*- "
KEYS = ["AMINO20", "DNA", "DNA5", "SEB8", "DAYHOFF", "KETO", "SEB14", "LIA10"]
batches = st.lists(st.text(alphabet=ALPHA, min_size=0, max_size=40), min_size=0, max_size=12)


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(seqs=batches, key=st.sampled_from(KEYS), eos=st.booleans(), bos=st.booleans(), pad=st.booleans(),
       extra=st.integers(0, 5), d=st.sampled_from("bhiqfd"))
def test_onehot_token_relationship_and_padding(oracle, seqs, key, eos, bos, pad, extra, d):
    tok = oracle.OracleTokenizer(key, eos, bos, pad)
    P = max([len(s) for s in seqs] + [0]) + eos + bos + extra
    if P <= 0:
        P = 1
    t = tok.batch_tokenize(seqs, padlen=P, destchar=d).astype(np.int64)      # (P, B)
    o = tok.batch_onehot_encode(seqs, padlen=P, destchar=d)                    # (P, B, C)
    assert o.shape == (P, len(seqs), tok.alphabet_size()) and set(np.unique(o)) <= {0, 1}
    rows = o.sum(axis=2)
    assert ((rows == 0) | (rows == 1)).all()
    hot = rows == 1
    assert (o.argmax(axis=2)[hot] == t[hot]).all()                             # SURVEY 8a-4 relationship
    assert (t[~hot] == 0).all()                                                # zero rows tokenise to 0
    assert np.array_equal(tok.batch_tokenize(seqs, padlen=P, destchar=d, batch_first=True).astype(np.int64), t.T)
    for i, s in enumerate(seqs):
        L = len(s.encode())
        if bos:
            assert t[0, i] == tok.bos()
        if eos:
            assert t[bos + L, i] == tok.eos()
        tail = t[bos + L + eos:, i]
        assert (tail == (tok.pad() if pad else 0)).all()
        assert hot[bos + L + eos:, i].all() == bool(pad) or tail.size == 0


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(seqs=batches, key=st.sampled_from(KEYS + ["BYTES"]), eos=st.booleans(), bos=st.booleans(), pad=st.booleans(),
       d=st.sampled_from("bhiqfd"), data=st.data())
def test_oracle_matches_compiled_reference(oracle, seqs, key, eos, bos, pad, d, data):
    ref = oracle.load_reference()
    if ref is None:
        pytest.skip("oracle/_ref not built (build container only)")
    r, o = ref.Tokenizer(key, eos, bos, pad), oracle.OracleTokenizer(key, eos, bos, pad)
    P = max([len(s) for s in seqs] + [0]) + eos + bos + data.draw(st.integers(0, 3))
    if P <= 0:
        P = 1
    mask = [np.array(data.draw(st.lists(st.integers(0, 1), min_size=len(s), max_size=len(s))), dtype=np.uint8)
            if data.draw(st.booleans()) else None for s in seqs]
    for bf in (False, True):
        a, b = r.batch_tokenize(seqs, padlen=P, destchar=d, batch_first=bf), o.batch_tokenize(seqs, padlen=P, destchar=d, batch_first=bf)
        assert a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes()
    a, b = r.batch_onehot_encode(seqs, padlen=P, destchar=d, mask=mask), o.batch_onehot_encode(seqs, padlen=P, destchar=d, mask=mask)
    assert a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes()
