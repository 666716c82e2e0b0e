"""k_tokens_pb8_fast's PAIRED TAIL (round 6): when padlen % 64 is 1 ... 32 the last position tile of sequence tile tb also carries the tail of
sequence tile tb + 8 (lanes whose piece would lie beyond padlen work on the partner's sequences) -- the (P,B) token matrix (the reference's
default layout, /root/reference/src/tokenize.h:420-425) and the raw-id pass of the two-pass one-hot (tokenize.h:326-330), against the oracle,
with the pairing on (automatic) and off (knob tokens_pb8_pair = 1), over the shapes where it applies and their neighbours where it must not."""
import ctypes

import numpy as np
import pytest

from bioseq_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[0, 1], ids=["paired", "unpaired"])
def pair_knob(request):
    from bioseq_amd import capi
    lib = capi.load()
    capi.check(lib.bsq_tuning_set(b"tokens_pb8_pair", request.param))
    yield request.param
    capi.check(lib.bsq_tuning_set(b"tokens_pb8_pair", 0))


@pytest.mark.parametrize("P", [160, 144, 65, 96, 97, 129, 288, 4112])
def test_seq_first_tokens_vs_oracle(gpu, bsq, oracle, pair_knob, P):
    """(P,B) int8 / int16 token matrices: B covers more than one group of eight 256-sequence tiles, with a ragged last tile, an odd number of
    groups (the last even group has no partner) and exactly one group (no pairing at all)"""
    import torch
    for B in (2048 + 256, 4096, 4096 + 2048 + 17 * 16, 20000, 2048):
        for key, flags in (("DNA4", (1, 1, 1)), ("AMINO20", (0, 0, 0)), ("PROTEIN", (1, 0, 1))):
            hi = P - flags[0] - flags[1]
            chars, offs = synth.synth_packed(1000 + B + P, B, max(0, hi - 40), hi, synth.DIRTY)
            tok, ora = bsq.Tokenizer(key, *flags), oracle.OracleTokenizer(key, *flags)
            dch, dof = torch.from_numpy(chars).to(gpu), torch.from_numpy(offs).to(gpu)
            for d in ("b", "h"):
                want = ora.tokenize_packed(chars, offs, P, d, False)
                got = tok.tokenize_packed(dch, dof, P, d, False).cpu().numpy()
                assert got.tobytes() == want.tobytes(), (P, B, key, flags, d)


@pytest.mark.parametrize("P,key,flags,destchar", [(160, "DNA4", (1, 1, 1), "B"), (160, "DNA4", (1, 1, 1), "f"), (96, "DNA5", (0, 0, 0), "B"),
                                                  (144, "AMINO20", (0, 0, 0), "f"), (272, "SEB8", (1, 1, 1), "B"), (161, "DNA4", (1, 1, 1), "f")])
def test_two_pass_onehot_raw_pass_vs_oracle(gpu, bsq, oracle, pair_knob, P, key, flags, destchar):
    """the raw-id pass (byte ids and nibble ids) of the two-pass one-hot, forced at sizes the oracle finishes quickly; and the slices of a
    large id matrix (two_pass_slice_mb = 1: the tail tile lies in the LAST slice only)"""
    import torch
    from bioseq_amd import capi
    lib = capi.load()
    B = 20000 + 48
    hi = P - flags[0] - flags[1]
    chars, offs = synth.synth_packed(77 + P, B, max(0, hi - 25), hi, "ACGTNacgtn" if key.startswith("DNA") else synth.DIRTY)
    tok, ora = bsq.Tokenizer(key, *flags), oracle.OracleTokenizer(key, *flags)
    want = ora.onehot_packed(chars, offs, P, destchar)
    dch, dof = torch.from_numpy(chars).to(gpu), torch.from_numpy(offs).to(gpu)
    capi.check(lib.bsq_tuning_set(b"onehot_path", 2))
    try:
        for nib, slice_mb in ((0, 0), (1, 0), (2, 0), (0, 1), (2, 1)):
            capi.check(lib.bsq_tuning_set(b"raw_nibbles", nib))
            capi.check(lib.bsq_tuning_set(b"two_pass_slice_mb", slice_mb))
            got = tok.onehot_packed(dch, dof, P, destchar).cpu().numpy()
            assert got.tobytes() == want.tobytes(), (P, key, destchar, nib, slice_mb)
    finally:
        for k in (b"onehot_path", b"raw_nibbles", b"two_pass_slice_mb"):
            capi.check(lib.bsq_tuning_set(k, 0))


def test_cfg4_full_size_paired_equals_unpaired(gpu):
    """BASELINE config 4 at full size (1M reads x 160): the int8 one-hot with the pairing on == off (both are checked against the reference's
    folds by bench.py and against its sha256 by test_cfg4_cfg5_full_size)"""
    import torch
    from bioseq_amd import capi
    lib = capi.load()
    c = synth.CONFIGS["cfg4"]
    chars, offs = synth.synth_packed(c["seed"], c["n"], c["lo"], c["hi"], c["letters"])
    desc = capi.make_desc(c["key"], c["eos"], c["bos"], c["padchar"])
    C = lib.bsq_alphabet_size(ctypes.byref(desc))
    dch, dof = torch.from_numpy(chars).to(gpu), torch.from_numpy(offs).to(gpu)
    outs = []
    for knob in (0, 1):
        capi.check(lib.bsq_tuning_set(b"tokens_pb8_pair", knob))
        out = torch.full((c["padlen"], c["n"], C), 3, dtype=torch.int8, device=gpu)
        capi.check(lib.bsq_onehot_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), None, c["n"], c["padlen"], capi.I8, out.data_ptr(), None))
        torch.cuda.synchronize()
        outs.append(out)
    capi.check(lib.bsq_tuning_set(b"tokens_pb8_pair", 0))
    assert torch.equal(outs[0], outs[1])
    assert int(outs[0].sum(dtype=torch.int64)) == c["padlen"] * c["n"]   # padchar: every position row holds exactly one 1
