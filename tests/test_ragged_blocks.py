"""GPU: column blocks of a wider (P, row_seqs, C) tensor through the RAGGED form of the two-pass stream (end of round 5): blocks that start
at any sequence of a tensor whose row pitch is no multiple of 4 KiB -- every position row of the block is cut at the chunk boundaries of
MEMORY, its first and last piece partial (EParams::ragged, bsq_onehot.hip).  Against the oracle's whole-batch encode: element types of
1 / 2 / 4 / 8 bytes, byte and nibble ids, LDS and LDS-free expansion, position slices, blocks written in any order, a block left out (its
bytes must stay untouched), a root that is itself off every boundary."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture
def knobs(bsq):
    from bioseq_amd import capi
    lib = capi.load()
    yield lib
    for k in (b"onehot_path", b"raw_nibbles", b"two_pass_slice_mb", b"expand_rows1"):
        capi.check(lib.bsq_tuning_set(k, 0))


CASES = [("DNA4", (1, 1, 1), "ACGT", "B", 5003, 130), ("DNA4", (1, 1, 1), "ACGT", "f", 3001, 70), ("DNA5", (0, 0, 0), "ACGTN", "B", 4100, 200),
         ("AMINO20", (0, 0, 0), "ACDEFGHIKLMNPQRSTVWY", "B", 2050, 129), ("AMINO20", (1, 1, 1), "ACDEFGHIKLMNPQRSTVWY", "f", 1500, 64),
         ("SEB8", (1, 0, 1), "ACDEFGHIKLMNPQRSTVWY", "h", 3000, 100), ("DNA4", (0, 0, 0), "ACGT", "d", 1111, 65), ("SEB14", (0, 0, 0), "ACDEFGHIKLMNPQRSTVWY", "B", 2500, 90)]


@pytest.mark.parametrize("key,flags,letters,dc,Bfull,P", CASES)
def test_ragged_column_blocks_equal_the_oracle(gpu, bsq, oracle, knobs, key, flags, letters, dc, Bfull, P):
    import torch
    from bioseq_amd import capi, synth
    lib = knobs
    ora = oracle.OracleTokenizer(key, *flags)
    chars, offs = synth.synth_packed(17 + Bfull, Bfull, 0, P - flags[0] - flags[1], letters)
    want = ora.onehot_packed(chars, offs, P, dc)
    sz, C = want.dtype.itemsize, want.shape[2]
    nbytes = want.size * sz
    desc = capi.make_desc(key, *flags)
    dt = ctypes.c_int(0)
    capi.check(lib.bsq_dtype_from_destchar(dc.encode(), ctypes.byref(dt)))
    dch, dof = torch.from_numpy(chars).to(gpu), torch.from_numpy(offs).to(gpu)
    rng = np.random.default_rng(Bfull)
    cuts = sorted(set([0, Bfull] + [int(x) for x in rng.integers(1, Bfull, 4)] + [Bfull // 2, Bfull // 2 + 1]))
    blocks = list(zip(cuts[:-1], cuts[1:]))
    capi.check(lib.bsq_tuning_set(b"onehot_path", 2))   # the stream form whatever the size
    want_b = np.frombuffer(want.tobytes(), dtype=np.uint8)
    for shift, nib, mb, skip in ((0, 0, 0, None), (8 * sz, 2, 0, 1), (4096 - sz, 1, 1, None), (sz, 2, 1, 2)):
        capi.check(lib.bsq_tuning_set(b"raw_nibbles", nib))
        capi.check(lib.bsq_tuning_set(b"two_pass_slice_mb", mb))
        buf = torch.full((nbytes + 8192,), 0x5A, dtype=torch.uint8, device=gpu)
        base = (-buf.data_ptr()) % 4096 + shift
        order = list(range(len(blocks)))
        rng.shuffle(order)
        for bi in order:
            if bi == skip:
                continue
            b0, b1 = blocks[bi]
            o = offs[b0:b1 + 1]
            sub = torch.from_numpy((o - o[0]).copy()).to(gpu)
            capi.check(lib.bsq_onehot_block_device(ctypes.byref(desc), dch.data_ptr() + int(o[0]), sub.data_ptr(), None, b1 - b0, P, dt,
                                                   buf.data_ptr() + base + b0 * C * sz, Bfull, None))
        torch.cuda.synchronize()
        h = buf.cpu().numpy()
        exp = want_b.copy().reshape(P, Bfull * C * sz)
        if skip is not None and skip < len(blocks):
            b0, b1 = blocks[skip]
            exp[:, b0 * C * sz:b1 * C * sz] = 0x5A   # the block that was left out: nobody may have written there
        assert h[base:base + nbytes].tobytes() == exp.tobytes(), (key, dc, Bfull, P, shift, nib, mb, skip, blocks)
        assert (h[:base] == 0x5A).all() and (h[base + nbytes:] == 0x5A).all(), (key, dc, shift, "wrote outside the root")
