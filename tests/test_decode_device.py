"""SURVEY.md section 8 row f-4 on the device: `decode_tokens` of token matrices that live in HBM, `decode_logits`
(argmax + decode, README.md:48) and the single-sequence `onehot_encode(device=...)` -- against the README vector and
against the host implementations of the same calls (which restate /root/reference/src/tokenize.h:131-216)."""
import itertools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ["DNA", "AMINO20", "SEB8", "DNA5", "BYTES"]
COMBOS = list(itertools.product([0, 1], repeat=3))


def valid_tokens(tok, rng, shape):
    """Random tokens drawn from everything the tokenizer can decode (letters' group ids and the special ids)."""
    ids = sorted(k for k in tok.token_decoder().keys() if 0 <= k < 128)   # (lut() of BYTES holds non-UTF-8 strings)
    ids += [i for i, on in ((tok.bos(), tok.includes_bos()), (tok.eos(), tok.includes_eos()), (tok.pad(), tok.is_padded())) if on]
    return rng.choice(np.array(ids, dtype=np.int64), size=shape)


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


@pytest.mark.parametrize("key", KEYS)
def test_device_decode_equals_host_decode(gpu, bsq, key):
    """1-D and 2-D, every item size, contiguous and strided views, ragged widths (<BOS>/<EOS>/<PAD> are 5 bytes)."""
    import torch
    rng = np.random.default_rng(7)
    for eos, bos, pad in COMBOS:
        tok = bsq.Tokenizer(key, eos, bos, pad)
        for shape in ((1, 1), (3, 64), (5, 65), (17, 200), (300, 33), (2, 1500)):
            toks = valid_tokens(tok, rng, shape)
            for np_dt, t_dt in ((np.uint8, torch.uint8), (np.int16, torch.int16), (np.int32, torch.int32), (np.int64, torch.int64)):
                if toks.max() > np.iinfo(np_dt).max:
                    continue
                host = toks.astype(np_dt)
                dev = torch.from_numpy(host).to(gpu)
                want = tok.decode_tokens(host)
                assert tok.decode_tokens(dev) == want
                assert tok.decode_tokens(dev[0]) == want[0]                      # 1-D
                assert tok.decode_tokens(dev.T) == tok.decode_tokens(np.ascontiguousarray(host.T))   # strided rows
                assert tok.decode_tokens(dev[:, ::2]) == tok.decode_tokens(host[:, ::2])              # strided columns
            if toks.max() <= 127:
                assert tok.decode_tokens(torch.from_numpy(toks.astype(np.int8)).to(gpu)) == tok.decode_tokens(toks.astype(np.int8))


def test_invalid_tokens_raise_like_the_host_path(gpu, bsq):
    import torch
    tok = bsq.Tokenizer("DNA", 1, 1, 1)   # ids 0..6
    for np_dt in (np.uint8, np.int16, np.int32, np.int64):
        host = np.zeros((4, 50), dtype=np_dt)
        host[2, 17] = 99
        host[3, 1] = 77          # not the first one in row-major order
        with pytest.raises(RuntimeError) as h:
            tok.decode_tokens(host)
        with pytest.raises(RuntimeError) as d:
            tok.decode_tokens(torch.from_numpy(host).to(gpu))
        assert str(d.value) == str(h.value) == "Unexpected/invalid token 99"
    with pytest.raises(ValueError, match="1 or 2 dimensions"):
        tok.decode_tokens(torch.zeros((2, 2, 2), dtype=torch.int8, device=gpu))
    assert tok.decode_tokens(torch.zeros((0, 5), dtype=torch.int8, device=gpu)) == []
    assert tok.decode_tokens(torch.zeros((3, 0), dtype=torch.int8, device=gpu)) == ["", "", ""]


@pytest.mark.parametrize("dtype", ["float32", "float16", "bfloat16", "float64"])
def test_decode_logits_equals_argmax_then_decode(gpu, bsq, dtype):
    import torch
    tok = bsq.Tokenizer("AMINO20", 1, 1, 1)
    C = tok.alphabet_size()
    g = torch.Generator(device="cpu").manual_seed(3)
    logits = torch.randn((9, 130, C), generator=g).to(getattr(torch, dtype)).to(gpu)
    logits[0, 0, 3] = logits[0, 0, 7] = 50.0          # a tie: the first maximum wins, as in torch.argmax
    want = tok.decode_tokens(logits.float().argmax(dim=2) if dtype != "float64" else logits.argmax(dim=2))
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


def test_single_sequence_onehot_on_device_equals_host(gpu, bsq):
    import torch
    for key, (eos, bos, pad) in itertools.product(("DNA", "AMINO20", "SEB8"), COMBOS):
        tok = bsq.Tokenizer(key, eos, bos, pad)
        for seq in ("", "A", "ACGTNXacgt*", "MKVLAAGIVGLLLAQ" * 9):
            for padlen in (0, len(seq), len(seq) + 1, len(seq) + 37):
                for dt in "BHIFDf":
                    want = tok.onehot_encode(seq, padlen, dt)
                    got = tok.onehot_encode(seq, padlen, dt, device="cuda")
                    assert got.is_cuda and tuple(got.shape) == want.shape, (key, eos, bos, pad, seq, padlen, dt)
                    assert got.cpu().numpy().tobytes() == want.tobytes(), (key, eos, bos, pad, seq, padlen, dt)
        assert (tok.onehot_encode(b"ACGT", 6, "f", device="cuda").cpu().numpy() == tok.onehot_encode(b"ACGT", 6, "f")).all()
        assert (tok.onehot_encode(bytearray(b"ACGT"), 6, "f", device=gpu).cpu().numpy() == tok.onehot_encode(bytearray(b"ACGT"), 6, "f")).all()
    with pytest.raises(RuntimeError, match="padlen is too short"):
        bsq.Tokenizer("DNA").onehot_encode("ACGT", 3, "f", device="cuda")
