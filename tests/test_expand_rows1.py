"""GPU: k_expand_rows1 -- the LDS-free expansion of one-byte one-hot rows of 3 ... 15 bytes (round 5) -- against the oracle, and against
k_expand_chunks on the same raw ids: every alphabet x flag combination whose row is 3 ... 15 bytes wide, batch sizes on and off the
id matrix's 256-sequence padding (chunks that run over the end of a position row), results that start off a 4-KiB / 16-byte
boundary (clipped and misaligned chunks: the byte loop), masks, ids handed in by the caller (bsq_onehot_from_raw_tokens_device) with
its own pitch and with no-token (255) entries, and a column block of a wider tensor."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ["DNA4", "DNA5", "KETO", "DAYHOFF", "SEB6", "SEB8", "SEB10", "SEB14", "LIA10", "MURPHY"]


@pytest.fixture
def two_pass(bsq):
    from bioseq_amd import capi
    lib = capi.load()
    capi.check(lib.bsq_tuning_set(b"onehot_path", 2))
    yield lib
    for k in (b"onehot_path", b"expand_rows1", b"raw_nibbles"):
        capi.check(lib.bsq_tuning_set(k, 0))


def test_every_small_row_width_vs_oracle(gpu, bsq, oracle, two_pass):
    import torch
    from bioseq_amd import capi, synth
    lib = two_pass
    seen = set()
    for ki, key in enumerate(KEYS):
        for flags in ((0, 0, 0), (1, 0, 0), (1, 1, 0), (1, 1, 1)):
            tok, ora = bsq.Tokenizer(key, *flags), oracle.OracleTokenizer(key, *flags)
            C = tok.alphabet_size()
            if not 3 <= C <= 15:
                continue
            seen.add(C)
            for B, P in ((700, 33), (256, 40), (1031, 17)):           # B % 256 != 0: the id matrix is padded, chunks wrap
                chars, offs = synth.synth_packed(100 * ki + B, B, 0, P - 2, synth.DIRTY)
                seqs = synth.unpack(chars, offs)
                exp = ora.batch_onehot_encode(seqs, padlen=P, destchar="B")
                dch, dof = torch.from_numpy(chars).to(gpu), torch.from_numpy(offs).to(gpu)
                # the new kernel on byte ids, on NIBBLE ids (k_expand_rows1<nibbles>), then k_expand_chunks on the same byte ids
                for rows1, nib in ((0, 1), (0, 2), (1, 1)):
                    capi.check(lib.bsq_tuning_set(b"expand_rows1", rows1))
                    capi.check(lib.bsq_tuning_set(b"raw_nibbles", nib))
                    got = tok.onehot_packed(dch, dof, P, "B")
                    assert got.cpu().numpy().tobytes() == exp.tobytes(), (key, flags, B, P, rows1, nib)
                capi.check(lib.bsq_tuning_set(b"expand_rows1", 0))
                # a result 16 / 1 / 4090 bytes off a 4-KiB boundary, guard bytes around it (C ABI on raw pointers), byte and nibble ids
                desc = capi.make_desc(key, *flags)
                for shift, nib in ((16, 1), (1, 1), (4090, 1), (16, 2), (1, 2), (4090, 2)):
                    capi.check(lib.bsq_tuning_set(b"raw_nibbles", nib))
                    buf = torch.full((exp.size + 8192,), 0x5A, dtype=torch.uint8, device=gpu)
                    base = (-buf.data_ptr()) % 4096 + shift
                    capi.check(lib.bsq_onehot_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), None, B, P, capi.I8,
                                                     buf.data_ptr() + base, None))
                    h = buf.cpu().numpy()
                    assert h[base:base + exp.size].tobytes() == exp.tobytes(), (key, flags, B, P, shift, nib)
                    assert (h[:base] == 0x5A).all() and (h[base + exp.size:] == 0x5A).all(), (key, flags, shift, nib)
                capi.check(lib.bsq_tuning_set(b"raw_nibbles", 0))
    assert seen >= {4, 5, 6, 7, 8, 9, 10, 11, 13, 14}, seen


def test_masks_and_larger_batches_vs_oracle(gpu, bsq, oracle, two_pass):
    import torch
    from bioseq_amd import synth
    for key, flags, B, P in (("DNA4", (1, 1, 1), 40000, 96), ("SEB8", (0, 1, 1), 9999, 300), ("DNA5", (0, 0, 0), 70001, 64)):
        chars, offs = synth.synth_packed(B, B, 0, P - 2, synth.DIRTY)
        seqs = synth.unpack(chars, offs)
        tok, ora = bsq.Tokenizer(key, *flags), oracle.OracleTokenizer(key, *flags)
        mask_b = (np.arange(chars.size) % 7 != 0).astype(np.uint8)
        mask = [mask_b[offs[i]:offs[i + 1]].copy() for i in range(B)]
        for m, mb in ((None, None), (mask, mask_b)):
            exp = ora.batch_onehot_encode(seqs, padlen=P, destchar="B", mask=m)
            got = tok.onehot_packed(torch.from_numpy(chars).to(gpu), torch.from_numpy(offs).to(gpu), P, "B",
                                    mask=None if mb is None else torch.from_numpy(mb).to(gpu))
            assert got.cpu().numpy().tobytes() == exp.tobytes(), (key, B, P, m is not None)


def test_caller_made_ids_with_their_own_pitch(gpu, bsq):
    """bsq_onehot_from_raw_tokens_device: a (P, pitch) id matrix of the caller's, 255 = no one, any pitch >= B."""
    import torch
    from bioseq_amd import capi
    lib = capi.load()
    rng = np.random.default_rng(5)
    for C in (3, 4, 7, 8, 12, 15):
        for B, P, pitch in ((1000, 21, 1000), (1000, 21, 1003), (4096, 9, 4096), (777, 50, 1024)):
            ids = rng.integers(0, C, size=(P, pitch)).astype(np.uint8)
            ids[rng.random(size=ids.shape) < 0.1] = 255
            exp = (ids[:, :B, None] == np.arange(C, dtype=np.uint8)[None, None, :]).astype(np.int8)
            d_ids = torch.from_numpy(ids).to(gpu)
            for rows1 in (0, 1):
                capi.check(lib.bsq_tuning_set(b"expand_rows1", rows1))
                try:
                    out = torch.full((P, B, C), 3, dtype=torch.int8, device=gpu)
                    capi.check(lib.bsq_onehot_from_raw_tokens_device(d_ids.data_ptr(), pitch, B, P, C, capi.I8, out.data_ptr(), None))
                    assert out.cpu().numpy().tobytes() == exp.tobytes(), (C, B, P, pitch, rows1)
                finally:
                    capi.check(lib.bsq_tuning_set(b"expand_rows1", 0))


def test_column_blocks_of_a_wider_int8_tensor(gpu, bsq, oracle):
    """bsq_onehot_block_device with 8-byte rows (DNA5 + BOS / EOS / PAD): blocks of 4096-sequence multiples are whole 4-KiB chunks per
    position row and go through the two-pass stream with a row gap -- the expansion is k_expand_rows1."""
    import torch
    from bioseq_amd import capi, synth
    lib = capi.load()
    key, flags, P, n_full = "DNA5", (1, 1, 1), 72, 3 * 8192
    chars, offs = synth.synth_packed(31, n_full, 0, P - 2, "ACGTN")
    seqs = synth.unpack(chars, offs)
    ora = oracle.OracleTokenizer(key, *flags)
    exp = ora.batch_onehot_encode(seqs, padlen=P, destchar="B")
    desc = capi.make_desc(key, *flags)
    C = exp.shape[2]
    assert C == 8
    root = torch.full((P, n_full, C), 9, dtype=torch.int8, device=gpu)
    dch, dof = torch.from_numpy(chars).to(gpu), torch.from_numpy(offs).to(gpu)
    for b0 in (8192, 0, 16384):   # out of order: a block that wrote into a neighbour would be caught
        sub = (dof[b0:b0 + 8192 + 1]).contiguous()
        capi.check(lib.bsq_onehot_block_device(ctypes.byref(desc), dch.data_ptr(), sub.data_ptr(), None, 8192, P, capi.I8,
                                               root.data_ptr() + b0 * C, n_full, None))
    assert root.cpu().numpy().tobytes() == exp.tobytes()
