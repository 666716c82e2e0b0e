"""GPU: the id scratch of the two-pass one-hot as NIBBLES (round 5: alphabets of at most 15 classes, elements of 2 bytes and more --
k_tokens_pb8_fast<nibbles> + k_expand_chunks reading two ids per byte) and the two-pass one-hot in SLICES of position rows (one scratch of a
slice's size, raw pass and expansion slice after slice) against the ORACLE, and against the byte-id / one-slice forms of the same call
(knobs raw_nibbles = 2 / 1, two_pass_slice_mb = 1 / -1).  Shapes: odd and even batch sizes (the last byte of a row holds one id), batches that are not a multiple of the
256-sequence tile, padlens that are not a multiple of the 64-position tile, results off a 4-KiB boundary (clipped first / last chunks),
sequences of length 0 and of exactly padlen - bos - eos, every flag combination, rows of 24 ... 63 bytes (the gated expansion) and
larger, column blocks of a wider tensor."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = [("DNA4", "ACGT"), ("DNA5", "ACGTN"), ("DAYHOFF", "ACDEFGHIKLMNPQRSTVWY"), ("SEB8", "ACDEFGHIKLMNPQRSTVWY"), ("SEB14", "ACDEFGHIKLMNPQRSTVWY"),
        ("LIA10", "ACDEFGHIKLMNPQRSTVWY")]


@pytest.fixture()
def knobs(gpu):
    from bioseq_amd import capi
    lib = capi.load()
    yield lib
    for k in (b"onehot_path", b"raw_nibbles", b"two_pass_slice_mb"):
        capi.check(lib.bsq_tuning_set(k, 0))


@pytest.mark.parametrize("key,letters", KEYS)
@pytest.mark.parametrize("flags", [(0, 0, 0), (1, 1, 1), (1, 0, 0), (0, 1, 1)])
def test_nibble_scratch_equals_the_oracle(gpu, bsq, oracle, knobs, key, letters, flags):
    import torch
    from bioseq_amd import capi, synth
    lib = knobs
    ora = oracle.OracleTokenizer(key, *flags)
    # (SEB14 with BOS / EOS / PAD has 17 classes: the library keeps byte ids there by itself -- both arms then run the same code)
    dev = torch.device("cuda:0")
    room = lambda P: P - flags[0] - flags[1]
    for si, (B, P, dc, shift) in enumerate([(1000, 70, "f", 0), (777, 129, "h", 0), (4097, 64, "f", 16), (513, 200, "i", 0), (300, 257, "d", 8),
                                            (2, 64, "f", 0), (1, 300, "f", 0), (65, 96, "H", 2)]):
        hi = room(P)
        chars, offs = synth.synth_packed(4000 + si, B, 0, hi, letters + letters.lower() + "*-")  # lengths 0 ... the whole room
        want = ora.onehot_packed(chars, offs, P, dc)
        sz = want.dtype.itemsize
        nbytes = want.size * sz
        buf = torch.full((nbytes + 4096 + 64,), 0x5A, dtype=torch.uint8, device=dev)
        base = (-buf.data_ptr()) % 4096 + shift * sz if shift else (-buf.data_ptr()) % 4096
        out = buf[base:base + nbytes]
        dch, dof = torch.from_numpy(chars).to(dev), torch.from_numpy(offs).to(dev)
        desc = capi.make_desc(key, *flags)
        dt = ctypes.c_int(0)
        capi.check(lib.bsq_dtype_from_destchar(dc.encode(), ctypes.byref(dt)))
        capi.check(lib.bsq_tuning_set(b"onehot_path", 2))
        got = {}
        for nib in (2, 1):   # nibbles wherever they apply / never
            capi.check(lib.bsq_tuning_set(b"raw_nibbles", nib))
            buf.fill_(0x5A)
            capi.check(lib.bsq_onehot_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), None, B, P, dt, out.data_ptr(), None))
            torch.cuda.synchronize()
            got[nib] = out.cpu().numpy().tobytes()
            assert bytes(buf[:base].cpu().numpy()) == b"\x5a" * base and bytes(buf[base + nbytes:].cpu().numpy()) == b"\x5a" * (len(buf) - base - nbytes), \
                (key, flags, B, P, dc, nib, "wrote outside the result")
        assert got[2] == want.tobytes(), (key, flags, B, P, dc, "nibble scratch differs from the oracle")
        assert got[1] == want.tobytes(), (key, flags, B, P, dc, "byte scratch differs from the oracle")


@pytest.mark.parametrize("dc,cuts", [("f", ((0, 3072), (3072, 4096), (4096, 8192))),
                                     # int8: 7-byte rows, blocks on 4096-sequence boundaries are whole chunks -> the gap stream + k_expand_rows1
                                     ("B", ((4096, 8192), (0, 4096))), ("B", ((0, 3072), (3072, 4096), (4096, 8192)))])
def test_nibble_scratch_column_blocks(gpu, bsq, oracle, knobs, dc, cuts):
    """A column block of a wider (P, row_seqs, C) tensor (a rank's shard stored into the root's buffer; a piece of a host batch) through the
    two-pass stream with a row gap: the nibble scratch against the oracle's whole-batch encode."""
    import torch
    from bioseq_amd import capi, synth
    lib = knobs
    key, flags, P = "DNA4", (1, 1, 1), 640   # ten position tiles: 1-MB slices cut the 3072- and 4096-sequence blocks in two / three
    ora = oracle.OracleTokenizer(key, *flags)
    dev = torch.device("cuda:0")
    Bfull = 8192  # pitch = 8192 * 7 * 4 bytes = 56 chunks: blocks on chunk boundaries every 1024 sequences
    chars, offs = synth.synth_packed(91, Bfull, 100, 638, "ACGT")
    want = ora.onehot_packed(chars, offs, P, dc)
    root = torch.zeros(want.shape, dtype=torch.float32 if dc == "f" else torch.int8, device=dev)
    sz = want.dtype.itemsize
    desc = capi.make_desc(key, *flags)
    dt = ctypes.c_int(0)
    capi.check(lib.bsq_dtype_from_destchar(dc.encode(), ctypes.byref(dt)))
    dch, dof = torch.from_numpy(chars).to(dev), torch.from_numpy(offs).to(dev)
    C = want.shape[2]
    capi.check(lib.bsq_tuning_set(b"onehot_path", 2))
    for nib, mb in ((2, -1), (1, -1), (2, 1), (1, 1)):
        capi.check(lib.bsq_tuning_set(b"raw_nibbles", nib))
        capi.check(lib.bsq_tuning_set(b"two_pass_slice_mb", mb))
        root.zero_()
        for b0, b1 in cuts:
            o = offs[b0:b1 + 1]
            sub_off = torch.from_numpy((o - o[0]).copy()).to(dev)
            capi.check(lib.bsq_onehot_block_device(ctypes.byref(desc), dch.data_ptr() + int(o[0]), sub_off.data_ptr(), None, b1 - b0, P, dt,
                                                   root.data_ptr() + b0 * C * sz, Bfull, None))
        torch.cuda.synchronize()
        assert root.cpu().numpy().tobytes() == want.tobytes(), ("column blocks", dc, nib, mb)


@pytest.mark.parametrize("key,flags,letters,dc", [("DNA4", (1, 1, 1), "ACGT", "f"), ("DNA5", (0, 0, 0), "ACGTN", "h"), ("AMINO20", (0, 0, 0), "ACDEFGHIKLMNPQRSTVWY", "B"),
                                                  ("AMINO20", (1, 1, 1), "ACDEFGHIKLMNPQRSTVWY", "f"), ("SEB8", (1, 0, 1), "ACDEFGHIKLMNPQRSTVWY", "d"),
                                                  # one-byte rows of 14 / 7 bytes: the sliced expansion is k_expand_rows1 (byte ids)
                                                  ("SEB14", (0, 0, 0), "ACDEFGHIKLMNPQRSTVWY", "B"), ("DNA4", (1, 1, 1), "ACGT", "B")])
def test_two_pass_slices_equal_the_oracle(gpu, bsq, oracle, knobs, key, flags, letters, dc):
    """two_pass_slice_mb = 1: every shape below is cut into 2 ... 16 slices of whole 64-position tiles (the last one ragged: padlen % 64 != 0),
    results on and off a 4-KiB boundary (a slice then starts inside a chunk: both neighbours write their part of it)."""
    import torch
    from bioseq_amd import capi, synth
    lib = knobs
    ora = oracle.OracleTokenizer(key, *flags)
    dev = torch.device("cuda:0")
    # (results with thin position slices -- fewer than four tiles -- are cut by SEQUENCES into column blocks instead: the first, second and
    #  fourth shape, and the third, which is misaligned: its blocks take the ragged form; the fifth has fat slices: four of eight tiles)
    for si, (B, P, shift) in enumerate([(20000, 300, 0), (9001, 257, 0), (5000, 1000, 3), (70000, 129, 0), (2000, 2048, 0), (50000, 48, 0)]):   # (the last: ONE position tile, too large all the same -- sequence blocks)
        hi = P - flags[0] - flags[1]
        chars, offs = synth.synth_packed(5000 + si, B, 0, hi, letters)
        want = ora.onehot_packed(chars, offs, P, dc)
        sz = want.dtype.itemsize
        nbytes = want.size * sz
        buf = torch.full((nbytes + 8192,), 0x5A, dtype=torch.uint8, device=dev)
        base = (-buf.data_ptr()) % 4096 + shift * sz
        out = buf[base:base + nbytes]
        dch, dof = torch.from_numpy(chars).to(dev), torch.from_numpy(offs).to(dev)
        desc = capi.make_desc(key, *flags)
        dt = ctypes.c_int(0)
        capi.check(lib.bsq_dtype_from_destchar(dc.encode(), ctypes.byref(dt)))
        capi.check(lib.bsq_tuning_set(b"onehot_path", 2))
        for nib, mb in ((0, 1), (2, 1), (1, 1), (0, -1)):
            capi.check(lib.bsq_tuning_set(b"raw_nibbles", nib))
            capi.check(lib.bsq_tuning_set(b"two_pass_slice_mb", mb))
            buf.fill_(0x5A)
            capi.check(lib.bsq_onehot_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), None, B, P, dt, out.data_ptr(), None))
            torch.cuda.synchronize()
            assert out.cpu().numpy().tobytes() == want.tobytes(), (key, flags, B, P, dc, nib, mb)
            h = buf.cpu().numpy()
            assert (h[:base] == 0x5A).all() and (h[base + nbytes:] == 0x5A).all(), (key, flags, B, P, nib, mb, "wrote outside the result")
        del buf, out
