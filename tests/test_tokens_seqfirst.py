"""The (P,B) int8 token matrix -- batch_tokenize's DEFAULT layout (batch_first=False, destchar 'B';
/root/reference/src/tokenize.cpp:82-98, tokenize.h:454-479) -- from k_tokens_pb8_fast (round 3: register-transposed tiles of
256 x 64, register or LDS alphabet table; 16-byte aligned rows) and from k_tokens_raw in value mode (every
other case; both of its tile shapes), and as the expansion scratch of the two-pass one-hot.  Bit-exact against the oracle."""
import ctypes
import itertools

import numpy as np
import pytest

from bioseq_amd import synth
from test_tokens8 import nasty_batch

pytestmark = pytest.mark.gpu

COMBOS = list(itertools.product([0, 1], repeat=3))  # (eos, bos, padchar)


# the quad-transposed kernel with each of its tiles, walk lengths and lookups (shapes it does not take fall to
# k_tokens_raw by themselves), then k_tokens_raw alone with each of ITS tiles
VARIANTS = {
    "auto": {},
    "pb8-ldslut": {"tokens8_lookup": 1},
    "pb8-aligned-only": {"tokens_pb8": 3},
    "raw-256x64": {"tokens_pb8": 1, "raw_mode": 1},
    "raw-1024x16": {"tokens_pb8": 1, "raw_mode": 4},
}


@pytest.fixture(params=list(VARIANTS), ids=list(VARIANTS))
def raw_mode(request):
    from bioseq_amd import capi
    lib = capi.load()
    for name, v in VARIANTS[request.param].items():
        capi.check(lib.bsq_tuning_set(name.encode(), v))
    yield request.param
    for name in ("tokens_pb8", "tokens8_lookup", "raw_mode"):
        capi.check(lib.bsq_tuning_set(name.encode(), 0))


def dev_tokens_pb(lib, capi, desc, chars, offs, P, gpu, shift=0, out_shift=0):
    import torch
    B = len(offs) - 1
    dch = torch.from_numpy(np.concatenate([np.zeros(shift, np.uint8), chars, np.full(1, 0x41, np.uint8)])).to(gpu)[shift:shift + len(chars)]
    dof = torch.from_numpy(offs).to(gpu)
    buf = torch.full((P * B + 32,), 99, dtype=torch.int8, device=gpu)
    out = buf[out_shift:out_shift + P * B].view(P, B)  # rows at 16- / 8- / 4- / 1-byte alignment
    capi.check(lib.bsq_tokenize_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), B, P, 0, capi.I8,
                                       out.data_ptr(), None))
    torch.cuda.synchronize()
    host = buf.cpu().numpy()
    assert (host[:out_shift] == 99).all() and (host[out_shift + P * B:] == 99).all(), "wrote outside the matrix"
    return host[out_shift:out_shift + P * B].reshape(P, B)


@pytest.mark.parametrize("B,lo,hi,P", [(1, 0, 0, 1), (1, 5, 5, 7), (3, 0, 9, 16), (1000, 1, 254, 256), (1004, 0, 62, 64), (1024, 0, 30, 33),
                                       (1025, 0, 14, 17), (4096, 0, 100, 100), (5000, 3, 60, 64), (9000, 0, 15, 15),
                                       (2047, 100, 300, 302), (70000, 0, 20, 24),
                                       # 16-byte aligned rows (k_tokens_pb8_fast): one piece, ragged tiles, padlen around 16 / 32 / 64
                                       (16, 0, 9, 11), (48, 0, 40, 41), (512, 0, 62, 64), (528, 10, 63, 65), (1040, 0, 127, 129),
                                       (4112, 0, 299, 301), (16400, 0, 30, 31), (272, 700, 1100, 1111),
                                       (48, 0, 69990, 70001), (17, 30000, 66000, 66001)])  # long padlen: > 1000 position tiles
def test_shapes_vs_oracle(gpu, oracle, raw_mode, B, lo, hi, P):
    """Single sequences, ragged tails of the sequence tile (B not a multiple of 16 / 256 / 1024), padlen that is not a
    multiple of the tile's 16 / 64 positions, empty sequences, every byte value; output rows at every alignment the
    kernel distinguishes (16-, 8-, 4-byte and byte stores)."""
    from bioseq_amd import capi
    lib = capi.load()
    hi = max(0, min(hi, P - 2))
    c2, o2 = nasty_batch(900 + B + P, B, min(lo, hi), hi)
    for key, (eos, bos, pad) in (("AMINO20", (0, 0, 0)), ("DNA", (1, 1, 1)), ("SEB8", (1, 0, 1)), ("DNA5", (0, 1, 0))):
        if hi + eos + bos > P:
            continue
        ora = oracle.OracleTokenizer(key, eos, bos, pad)
        want = ora.tokenize_packed(c2, o2, P, "b", False)
        for shift, out_shift in ((0, 0), (3, 8), (1, 4), (2, 1)):
            got = dev_tokens_pb(lib, capi, capi.make_desc(key, eos, bos, pad), c2, o2, P, gpu, shift, out_shift)
            assert got.shape == want.shape and got.tobytes() == want.tobytes(), (key, eos, bos, pad, B, P, shift, out_shift)


def test_cfg2_seq_first_digest(gpu, raw_mode):
    """BASELINE cfg2 seq-first at full size against the reference's digest."""
    import hashlib
    from bioseq_amd import capi
    lib = capi.load()
    chars, offs = synth.synth_packed(202, 65536, 50, 1024, synth.AA)
    got = dev_tokens_pb(lib, capi, capi.make_desc("AMINO20", 0, 0, 0), chars, offs, 1024, gpu)
    assert int(got.sum(dtype=np.int64)) == 333614835
    assert hashlib.sha256(got.tobytes()).hexdigest() == "9d6b9328e8e6605dff897d6ebec8a1f77b81f69ed666bd9b5a59e5f750cd1b73"


def test_two_pass_onehot_with_either_scratch_tile(gpu, oracle, raw_mode):
    """The expansion scratch written by the wide tile expands to the same one-hot."""
    import torch
    from bioseq_amd import capi
    lib = capi.load()
    capi.check(lib.bsq_tuning_set(b"onehot_path", 2))
    try:
        B, P = 3000, 70
        chars, offs = nasty_batch(77, B, 0, P - 2)
        for key, (eos, bos, pad) in (("AMINO20", (0, 0, 0)), ("DNA", (1, 1, 1))):
            desc = capi.make_desc(key, eos, bos, pad)
            C = lib.bsq_alphabet_size(ctypes.byref(desc))
            dch, dof = torch.from_numpy(chars).to(gpu), torch.from_numpy(offs).to(gpu)
            out = torch.full((P, B, C), 9, dtype=torch.float32, device=gpu)
            capi.check(lib.bsq_onehot_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), None, B, P, capi.F32,
                                             out.data_ptr(), None))
            torch.cuda.synchronize()
            want = oracle.OracleTokenizer(key, eos, bos, pad).onehot_packed(chars, offs, P, "f")
            assert out.cpu().numpy().tobytes() == want.tobytes(), key
    finally:
        capi.check(lib.bsq_tuning_set(b"onehot_path", 0))


@pytest.mark.parametrize("dc", list("hilfd"))
@pytest.mark.parametrize("B,P", [(1, 5), (7, 64), (65, 100), (1001, 33), (4099, 70), (20001, 24), (4096, 64)])
def test_wider_types_any_batch_size_any_alignment(gpu, oracle, dc, B, P):
    """(P,B) token matrices of 2- / 4- / 8-byte elements (k_tokenize_tile) when the rows are only element-aligned -- odd
    batch sizes, outputs offset by a few elements: segments cut at the output's 16-byte lines -- and in the round-1 form
    (knob tokenize_path = 2); every sequences-per-tile setting; nothing outside the matrix is written."""
    import torch
    from bioseq_amd import capi
    lib = capi.load()
    chars, offs = nasty_batch(B * 5 + P, B, 0, P - 2)
    dch = torch.from_numpy(np.concatenate([chars, np.zeros(1, np.uint8)])).to(gpu)
    dof = torch.from_numpy(offs).to(gpu)
    dt = ctypes.c_int(0)
    capi.check(lib.bsq_dtype_from_destchar(dc.encode(), ctypes.byref(dt)))
    sz = lib.bsq_dtype_size(dt)
    try:
        for key, flags in (("AMINO20", (0, 0, 0)), ("DNA", (1, 1, 1))):
            want = oracle.OracleTokenizer(key, *flags).tokenize_packed(chars, offs, P, dc, False)
            desc = capi.make_desc(key, *flags)
            # (tokens_pb8 = 3: k_tokenize_tile for every unaligned shape; 0: int16 through the unaligned-row form of k_tokens_pb8_fast)
            for path, tb, shift, pb8 in ((0, 0, 0, 3), (0, 0, 1, 3), (0, 64, 3, 3), (0, 256, 1, 3), (2, 0, 1, 3), (0, 128, 2, 3), (0, 0, 1, 0), (0, 0, 0, 0), (0, 0, 3, 0)):
                capi.check(lib.bsq_tuning_set(b"tokenize_path", path))
                capi.check(lib.bsq_tuning_set(b"tokenize_tb", tb))
                capi.check(lib.bsq_tuning_set(b"tokens_pb8", pb8))
                buf = torch.full(((P * B + 24) * sz,), 0x5A, dtype=torch.uint8, device=gpu)
                lo = shift * sz
                capi.check(lib.bsq_tokenize_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), B, P, 0, dt,
                                                   buf.data_ptr() + lo, None))
                torch.cuda.synchronize()
                host = buf.cpu().numpy()
                assert (host[:lo] == 0x5A).all() and (host[lo + P * B * sz:] == 0x5A).all(), "wrote outside the matrix"
                assert host[lo:lo + P * B * sz].tobytes() == want.tobytes(), (key, flags, dc, path, tb, shift, pb8)
    finally:
        capi.check(lib.bsq_tuning_set(b"tokenize_path", 0))
        capi.check(lib.bsq_tuning_set(b"tokenize_tb", 0))
        capi.check(lib.bsq_tuning_set(b"tokens_pb8", 0))


@pytest.mark.parametrize("dc", list("hilfd"))
@pytest.mark.parametrize("B,P", [(8, 5), (16, 64), (264, 100), (1000, 33), (4104, 70), (20000, 24), (4096, 129), (520, 1100)])
def test_wider_types_aligned_rows_through_pb8(gpu, oracle, dc, B, P):
    """(P,B) token matrices of 2- / 4- / 8-byte elements whose rows are 16-byte aligned: k_tokens_pb8_fast builds the tile in
    bytes and widens on the way out (register and LDS alphabet table; knob tokens_pb8 = 2 takes it for the 4- / 8-byte types
    too, where it is slower than k_tokenize_tile), against k_tokenize_tile (knob 1), the automatic choice and the oracle; nothing outside the matrix is written."""
    import torch
    from bioseq_amd import capi
    lib = capi.load()
    chars, offs = nasty_batch(B * 3 + P, B, 0, P - 2)
    dch = torch.from_numpy(np.concatenate([chars, np.zeros(1, np.uint8)])).to(gpu)
    dof = torch.from_numpy(offs).to(gpu)
    dt = ctypes.c_int(0)
    capi.check(lib.bsq_dtype_from_destchar(dc.encode(), ctypes.byref(dt)))
    sz = lib.bsq_dtype_size(dt)
    try:
        for key, flags in (("AMINO20", (0, 0, 0)), ("DNA", (1, 1, 1)), ("SEB8", (1, 0, 0))):
            want = oracle.OracleTokenizer(key, *flags).tokenize_packed(chars, offs, P, dc, False)
            desc = capi.make_desc(key, *flags)
            for pb8, lookup in ((2, 0), (2, 1), (0, 0), (1, 0)):
                capi.check(lib.bsq_tuning_set(b"tokens_pb8", pb8))
                capi.check(lib.bsq_tuning_set(b"tokens8_lookup", lookup))
                buf = torch.full(((P * B + 32) * sz,), 0x5A, dtype=torch.uint8, device=gpu)
                lo = 16
                capi.check(lib.bsq_tokenize_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), B, P, 0, dt,
                                                   buf.data_ptr() + lo, None))
                torch.cuda.synchronize()
                host = buf.cpu().numpy()
                assert (host[:lo] == 0x5A).all() and (host[lo + P * B * sz:] == 0x5A).all(), "wrote outside the matrix"
                assert host[lo:lo + P * B * sz].tobytes() == want.tobytes(), (key, flags, dc, pb8, lookup)
    finally:
        capi.check(lib.bsq_tuning_set(b"tokens_pb8", 0))
        capi.check(lib.bsq_tuning_set(b"tokens8_lookup", 0))
