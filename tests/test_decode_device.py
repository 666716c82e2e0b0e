"""SURVEY.md section 8 row f-4 on the device: `decode_tokens` of token matrices that live in HBM, `decode_logits`
(argmax + decode, README.md:48) and the single-sequence `onehot_encode(device=...)`.

Every expected value here is REFERENCE-DERIVED (tests/golden/make_golden.py, run in the build container against
oracle/_ref = the reference's own C++ compiled in place):
  * decode.json.gz + decode_tokens.npz -- what the reference's decode_tokens (src/tokenize.h:131-183) returns for stored
    token matrices (2-D, 1-D rows, transposed and column-strided views);
  * alphabets.json `lut` -- the reference's token -> piece table (`Tokenizer.lut()`), joined here for shapes the fixture
    does not hold;
  * single.json.gz -- dtype / shape / sha256 of the reference's single-sequence onehot_encode (src/tokenize.h:188-216,
    src/tokenize.cpp:10-51);
  * kats.json -- the README vector.
The CPU half of this file pins the product's HOST implementations of the same calls to the same fixtures."""
import gzip
import hashlib
import itertools
import json
import os

import numpy as np
import pytest

COMBOS = list(itertools.product([0, 1], repeat=3))
ITEM_TYPES = ("int8", "uint8", "int16", "int32", "int64")


@pytest.fixture(scope="module")
def decode_fixture(golden_dir):
    with gzip.open(os.path.join(golden_dir, "decode.json.gz")) as f:
        cases = json.load(f)
    return cases, np.load(os.path.join(golden_dir, "decode_tokens.npz"))


@pytest.fixture(scope="module")
def single_fixture(golden_dir):
    with gzip.open(os.path.join(golden_dir, "single.json.gz")) as f:
        rows = json.load(f)
    seqs = {r["key"]: r["seqs"] for r in rows if isinstance(r, dict)}
    return [r for r in rows if isinstance(r, list)], seqs


def sha24(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:24]


def reference_join(alphabets_golden, key, eos, bos, pad, toks):
    """Decode with the table the REFERENCE dumped (alphabets.json: Tokenizer.lut() per key x flags)."""
    lut = alphabets_golden["meta"][key][f"{eos}{bos}{pad}"]["lut"]
    return ["".join(lut[str(int(v))] for v in row) for row in toks]


def valid_ids(alphabets_golden, key, eos, bos, pad):
    m = alphabets_golden["meta"][key][f"{eos}{bos}{pad}"]
    return np.array(sorted(int(k) for k in m["lut"] if int(k) >= 0), dtype=np.int64)


# ---------------------------------------------------------------- CPU: host implementations vs the fixtures

def test_host_decode_matches_reference_fixture(bsq, decode_fixture):
    cases, arrays = decode_fixture
    assert len(cases) == 7 * 8 * 5
    for c in cases:
        tok = bsq.Tokenizer(c["key"], c["eos"], c["bos"], c["padchar"])
        toks = arrays[c["name"]]
        for dt in ITEM_TYPES + ("uint16", "uint32", "uint64"):
            if toks.max() <= np.iinfo(dt).max:
                assert tok.decode_tokens(toks.astype(dt)) == c["decoded"], (c["name"], dt)
        assert [tok.decode_tokens(toks[r]) for r in range(toks.shape[0])] == c["decoded"]
        assert tok.decode_tokens(toks.astype(np.int16).T) == c["transposed"]
        assert tok.decode_tokens(toks.astype(np.int32)[:, ::2]) == c["strided"]


def test_host_single_sequence_onehot_matches_reference_fixture(bsq, single_fixture):
    rows, seqs = single_fixture
    assert len(rows) > 5000
    for key, eos, bos, pad, si, padlen, kind, dt, dtype, shape, ones, digest in rows:
        tok = bsq.Tokenizer(key, eos, bos, pad)
        s = seqs[key][si]
        obj = s if kind == "str" else s.encode() if kind == "bytes" else bytearray(s.encode())
        a = tok.onehot_encode(obj, padlen, dt) if dt else tok.onehot_encode(obj, padlen)
        assert (str(a.dtype), list(a.shape), int(a.sum()), sha24(a)) == (dtype, shape, ones, digest), \
            (key, eos, bos, pad, si, padlen, kind, dt)


# ---------------------------------------------------------------- GPU

@pytest.mark.gpu
def test_readme_vector_on_device(gpu, bsq, kats):
    import torch
    tok = bsq.pbeos_tokenizers["DNA"]
    t = torch.tensor(kats["readme"]["tokens"], dtype=torch.int8, device=gpu)
    assert tok.decode_tokens(t) == kats["readme"]["decoded"]
    assert tok.decode_tokens(t[1]) == kats["readme"]["decoded"][1]
    assert tok.decode_tokens(t.to(torch.long)) == kats["readme"]["decoded"]
    # seq-first layout decoded through a transposed VIEW (no copy): strides are honoured on the device
    sf = tok.batch_tokenize(["ACGT", "GGGG"], padlen=7, device="cuda")
    assert tuple(sf.shape) == (7, 2) and tok.decode_tokens(sf.T) == kats["readme"]["decoded"]


@pytest.mark.gpu
def test_device_decode_matches_reference_fixture(gpu, bsq, decode_fixture):
    """Every fixture case: 2-D in every item size torch has, 1-D rows, transposed and column-strided device VIEWS."""
    import torch
    cases, arrays = decode_fixture
    for c in cases:
        tok = bsq.Tokenizer(c["key"], c["eos"], c["bos"], c["padchar"])
        toks = arrays[c["name"]]
        for dt in ITEM_TYPES:
            if toks.max() > np.iinfo(dt).max:
                continue
            dev = torch.from_numpy(toks.astype(dt)).to(gpu)
            assert tok.decode_tokens(dev) == c["decoded"], (c["name"], dt)
            assert tok.decode_tokens(dev.T) == c["transposed"], (c["name"], dt)
            assert tok.decode_tokens(dev[:, ::2]) == c["strided"], (c["name"], dt)
        dev = torch.from_numpy(toks.astype(np.int32)).to(gpu)
        for r in (0, toks.shape[0] - 1):
            assert tok.decode_tokens(dev[r]) == c["decoded"][r]                      # 1-D -> str


@pytest.mark.gpu
@pytest.mark.parametrize("key", ["DNA", "AMINO20", "SEB8", "DNA5", "KETO", "LIA10"])
def test_device_decode_large_shapes_vs_reference_table(gpu, bsq, alphabets_golden, key):
    """Shapes beyond the fixture (many rows, rows longer than a wave's 64 pieces x several rounds): expected strings are
    joined from the reference's own lut() dump."""
    import torch
    rng = np.random.default_rng(7)
    for eos, bos, pad in COMBOS:
        tok = bsq.Tokenizer(key, eos, bos, pad)
        ids = valid_ids(alphabets_golden, key, eos, bos, pad)
        for shape in ((1, 1), (300, 33), (2, 1500), (17, 4097)):
            toks = rng.choice(ids, size=shape)
            want = reference_join(alphabets_golden, key, eos, bos, pad, toks)
            for np_dt in (np.uint8, np.int16, np.int32, np.int64):
                dev = torch.from_numpy(toks.astype(np_dt)).to(gpu)
                assert tok.decode_tokens(dev) == want
            assert tok.decode_tokens(dev[0]) == want[0]
            assert tok.decode_tokens(dev.T) == reference_join(alphabets_golden, key, eos, bos, pad, toks.T)
            assert tok.decode_tokens(dev[:, ::3]) == reference_join(alphabets_golden, key, eos, bos, pad, toks[:, ::3])


@pytest.mark.gpu
def test_invalid_tokens_raise_with_the_reference_text(gpu, bsq):
    import torch
    tok = bsq.Tokenizer("DNA", 1, 1, 1)   # ids 0..6
    for np_dt in (np.uint8, np.int16, np.int32, np.int64):
        host = np.zeros((4, 50), dtype=np_dt)
        host[2, 17] = 99
        host[3, 1] = 77          # not the first one in row-major order
        with pytest.raises(RuntimeError) as d:
            tok.decode_tokens(torch.from_numpy(host).to(gpu))
        assert str(d.value) == "Unexpected/invalid token 99"     # tokenize.h:150,170
    with pytest.raises(ValueError, match="1 or 2 dimensions"):   # tokenize.h:139
        tok.decode_tokens(torch.zeros((2, 2, 2), dtype=torch.int8, device=gpu))
    assert tok.decode_tokens(torch.zeros((0, 5), dtype=torch.int8, device=gpu)) == []
    assert tok.decode_tokens(torch.zeros((3, 0), dtype=torch.int8, device=gpu)) == ["", "", ""]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float32", "float16", "bfloat16", "float64"])
def test_decode_logits_equals_argmax_then_reference_table(gpu, bsq, alphabets_golden, dtype):
    import torch
    tok = bsq.Tokenizer("AMINO20", 1, 1, 1)
    C = tok.alphabet_size()
    g = torch.Generator(device="cpu").manual_seed(3)
    logits = torch.randn((9, 130, C), generator=g).to(getattr(torch, dtype)).to(gpu)
    logits[0, 0, 3] = logits[0, 0, 7] = 50.0          # a tie: the first maximum wins, as in torch.argmax
    logits[1, 5, 9] = float("nan")                    # NaN is the maximum for torch.argmax
    logits[2, 0, 0] = float("nan")
    logits[2, 0, 4] = float("nan")                    # the first NaN wins
    am = (logits.float() if dtype != "float64" else logits).argmax(dim=2).cpu().numpy()
    assert am[0, 0] == 3 and am[1, 5] == 9 and am[2, 0] == 0
    want = reference_join(alphabets_golden, "AMINO20", 1, 1, 1, am)
    assert tok.decode_logits(logits) == want
    assert tok.decode_logits(logits[4]) == want[4]                                   # (L, C) -> str
    assert tok.decode_logits(logits.transpose(0, 1).contiguous(), batch_first=False) == want   # (L, B, C)
    assert tok.decode_logits(logits.transpose(0, 1), batch_first=False) == want       # non-contiguous input
    # end to end: the one-hot of a batch, taken as logits, decodes to the batch
    seqs = ["ACDEFGHIK", "LMNPQRSTVWY", ""]
    oh = tok.batch_onehot_encode(seqs, padlen=16, destchar="f", device="cuda")      # (P, B, C)
    dec = tok.decode_logits(oh, batch_first=False)
    assert [d.split("<EOS>")[0].replace("<BOS>", "") for d in dec] == seqs
    with pytest.raises(ValueError):
        tok.decode_logits(torch.zeros((4, C), dtype=torch.int32, device=gpu))


@pytest.mark.gpu
def test_single_sequence_onehot_on_device_matches_reference_fixture(gpu, bsq, single_fixture):
    import torch
    names = {"uint8": torch.uint8, "uint16": torch.int16, "uint32": torch.int32, "float32": torch.float32,
             "float64": torch.float64}   # device tensors of the unsigned 2/4-byte types carry the same bits as signed
    rows, seqs = single_fixture
    for key, eos, bos, pad, si, padlen, kind, dt, dtype, shape, ones, digest in rows:
        tok = bsq.Tokenizer(key, eos, bos, pad)
        s = seqs[key][si]
        obj = s if kind == "str" else s.encode() if kind == "bytes" else bytearray(s.encode())
        got = tok.onehot_encode(obj, padlen, dt, device="cuda") if dt else tok.onehot_encode(obj, padlen, device=gpu)
        assert got.is_cuda and got.dtype == names[dtype] and list(got.shape) == shape, (key, eos, bos, pad, si, padlen, kind, dt)
        assert sha24(got.cpu().numpy()) == digest, (key, eos, bos, pad, si, padlen, kind, dt)
    with pytest.raises(RuntimeError, match="padlen is too short"):     # tokenize.h:191
        bsq.Tokenizer("DNA").onehot_encode("ACGT", 3, "f", device="cuda")
