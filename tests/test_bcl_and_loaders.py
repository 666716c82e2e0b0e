"""Channels-first (B,C,P) one-hot written directly by the kernel, and the batch-granular dataset layer
(SURVEY.md 8f-3).  Expected values: the oracle's (P,B,C) tensor, transposed."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("key,flags,d", [("AMINO20", (0, 0, 0), "f"), ("DNA", (1, 1, 1), "b"), ("SEB8", (1, 0, 1), "d"),
                                          ("DNA5", (0, 1, 0), "h"), ("BYTES", (1, 1, 1), "i")])
def test_bcl_equals_transposed_reference_layout(gpu, bsq, oracle, key, flags, d):
    import torch
    from bioseq_amd import synth
    for B, P, hi in ((257, 64, 60), (33, 200, 198), (5, 17, 10), (1000, 16, 14)):
        chars, offs = synth.synth_packed(B + P, B, 0, hi, synth.DIRTY)
        seqs = synth.unpack(chars, offs)
        tok, ora = bsq.Tokenizer(key, *flags), oracle.OracleTokenizer(key, *flags)
        mask_b = (np.arange(chars.size) % 5 != 0).astype(np.uint8)
        mask = [mask_b[offs[i]:offs[i + 1]].copy() for i in range(B)]
        for m, mb in ((None, None), (mask, mask_b)):
            exp = np.ascontiguousarray(ora.batch_onehot_encode(seqs, padlen=P, destchar=d, mask=m).transpose(1, 2, 0))
            got = tok.batch_onehot_encode(seqs, padlen=P, destchar=d, mask=m, layout="bcl")
            assert got.dtype == exp.dtype and got.shape == exp.shape == (B, tok.alphabet_size(), P)
            assert got.tobytes() == exp.tobytes()
            dev = tok.onehot_packed(torch.from_numpy(chars).to(gpu), torch.from_numpy(offs).to(gpu), P, d,
                                    mask=None if mb is None else torch.from_numpy(mb).to(gpu), layout="bcl")
            assert dev.is_cuda and dev.cpu().numpy().tobytes() == exp.tobytes()
    with pytest.raises(ValueError, match="layout"):
        tok.batch_onehot_encode(seqs, padlen=P, layout="cbl")


def test_dataset_batches_match_per_item_reference_semantics(gpu, bsq, oracle, tmp_path):
    import torch
    from bioseq_amd import synth
    from bioseq_amd.flatfile import FlatFile, write_flatfile
    from bioseq_amd.loaders import FF2NP, AugmentedSeqDataset, FlatFileDataset
    chars, offs = synth.synth_packed(12, 500, 1, 120, synth.AA)
    seqs = synth.unpack(chars, offs)
    ff = FlatFile(write_flatfile(seqs, str(tmp_path / "d.ff")))
    tok, ora = bsq.pbeos_tokenizers["PROTEIN"], oracle.OracleTokenizer("PROTEIN", 1, 1, 1)
    P = ff.maxseqlen + 2
    exp_tok = ora.batch_tokenize(seqs, padlen=P, batch_first=True)
    exp_oh = np.ascontiguousarray(ora.batch_onehot_encode(seqs, padlen=P, destchar="f").transpose(1, 2, 0))
    ds = FlatFileDataset(ff, tok, device=gpu)
    assert len(ds) == 500 and ds.max_seq_len == P
    b = ds.get_batch(100, 228)
    assert b.dtype == torch.long and tuple(b.shape) == (128, P) and (b.cpu().numpy() == exp_tok[100:228]).all()
    assert (ds[7].cpu().numpy() == exp_tok[7]).all() and tuple(ds[7].shape) == (P,)           # reference item shape
    assert (ds.__getitems__([5, 499, 17]).cpu().numpy() == exp_tok[[5, 499, 17]]).all()       # scattered indices
    assert (ds[10:20].cpu().numpy() == exp_tok[10:20]).all()
    assert (ds[-1].cpu().numpy() == exp_tok[499]).all()                                        # Python indexing ...
    for bad in (500, -501, 10 ** 6):                                                           # ... never a silent wrap
        with pytest.raises(IndexError):
            ds[bad]
    with pytest.raises(IndexError):
        ds.__getitems__([1, 500])
    assert sum(1 for _ in ds) == 500                                                           # legacy iteration ends
    dl = torch.utils.data.DataLoader(ds, batch_size=64, shuffle=False, collate_fn=lambda x: x)
    got = torch.cat([x for x in dl]).cpu().numpy()
    assert (got == exp_tok).all()
    cnn = FlatFileDataset(ff, tok, cnn=True, device=gpu)
    o = cnn.get_batch(0, 500)
    assert o.dtype == torch.float32 and o.is_contiguous() and tuple(o.shape) == (500, tok.alphabet_size(), P)
    assert o.cpu().numpy().tobytes() == exp_oh.tobytes()
    assert cnn[3].cpu().numpy().tobytes() == exp_oh[3].tobytes()
    # augmentation: ~half of the sequences differ in exactly one residue; the resident store is untouched
    aug = AugmentedSeqDataset(ff, tok, device=gpu)
    a = aug.get_batch(0, 500).cpu().numpy()
    ndiff = (a != exp_tok).sum(axis=1)
    assert set(np.unique(ndiff)) <= {0, 1} and 180 < (ndiff == 1).sum() < 320
    assert (ds.get_batch(0, 500).cpu().numpy() == exp_tok).all()
    # int8 token rows: the loader's augmented batch step is the one-call entry (one launch); same mutations as the int64 rows
    # of a dataset with the same seed, the store untouched, shuffled index batches too
    aug8 = AugmentedSeqDataset(ff, tok, device=gpu, token_dtype="b")
    a8 = aug8.get_batch(0, 500)
    assert a8.dtype == torch.int8 and (a8.cpu().numpy() == a).all()
    idx = torch.randperm(500, device=gpu)[:200]
    aug64 = AugmentedSeqDataset(ff, tok, device=gpu)
    aug64.get_batch(0, 1)                                  # both datasets at call 2 (same seed stream)
    assert (aug8.__getitems__(idx).cpu().numpy() == aug64.__getitems__(idx).cpu().numpy()).all()
    from bioseq_amd import blosum
    blosum.check_fused(synchronize=True)
    assert (ds.get_batch(0, 500).cpu().numpy() == exp_tok).all()
    # FF2NP: token memmap of the whole store
    mat, path = FF2NP(ff, tok, str(tmp_path / "toks.u8"), batch_size=128)
    assert mat.shape == (500, P) and (np.asarray(mat).view(np.int8) == exp_tok).all()


@pytest.mark.parametrize("key,flags", [("AMINO20", (0, 0, 0)), ("DNA", (1, 1, 1)), ("SEB8", (1, 0, 1)), ("DNA5", (0, 1, 0))])
def test_bcl_two_pass_form_equals_single_pass_and_oracle(gpu, bsq, oracle, key, flags):
    """The two-pass channels-first path (raw (B,P) ids from k_tokens_bp8 + k_expand_bcl; automatic for outputs >= 256 MB,
    forced here with the knob `bcl_path`) against the single-pass kernel and the transposed oracle: every element type,
    padlens of 128 and more that are multiples of 16, dirty input (every kind of unmapped byte -> all-zero column), with
    and without a mask."""
    import torch
    from bioseq_amd import capi, synth
    lib = capi.load()
    tok, ora = bsq.Tokenizer(key, *flags), oracle.OracleTokenizer(key, *flags)
    for B, P in ((300, 128), (77, 272), (5, 1024), (1, 144)):
        chars, offs = synth.synth_packed(B * 3 + P, B, 0, P - 2, synth.DIRTY)
        dch, dof = torch.from_numpy(chars).to(gpu), torch.from_numpy(offs).to(gpu)
        mask = (np.random.default_rng(B + P).random(chars.size) < 0.7).astype(np.uint8)
        dm = torch.from_numpy(mask).to(gpu)
        for d in "bhifd":
            for m, mdev in ((None, None), (mask, dm)):  # masked: k_tokens_bp8<.., MASK> in raw-id mode
                exp = np.ascontiguousarray(ora.onehot_packed(chars, offs, P, d, mask=m).transpose(1, 2, 0))
                got = {}
                for path in (1, 2):
                    capi.check(lib.bsq_tuning_set(b"bcl_path", path))
                    try:
                        got[path] = tok.onehot_packed(dch, dof, P, d, mask=mdev, layout="bcl").cpu().numpy()
                    finally:
                        capi.check(lib.bsq_tuning_set(b"bcl_path", 0))
                assert got[2].tobytes() == exp.tobytes(), (B, P, d, m is not None)
                assert got[1].tobytes() == exp.tobytes(), (B, P, d, m is not None)


@pytest.mark.parametrize("key,flags", [("AMINO20", (0, 0, 0)), ("DNA", (1, 1, 1))])
def test_bcl_two_pass_in_slices_of_sequences(gpu, bsq, oracle, key, flags):
    """Round 5: the two-pass channels-first path in SLICES of sequences (one id scratch of a slice's size; automatic beyond 128 MB of ids,
    forced here with two_pass_slice_mb = 1: slices of 2048 / 1792 / 512 sequences, the last one ragged) against the transposed oracle,
    with and without a mask, int8 and float32."""
    import torch
    from bioseq_amd import capi, synth
    lib = capi.load()
    tok, ora = bsq.Tokenizer(key, *flags), oracle.OracleTokenizer(key, *flags)
    try:
        capi.check(lib.bsq_tuning_set(b"bcl_path", 2))
        for B, P in ((5000, 512), (4000, 576), (1300, 2048)):
            chars, offs = synth.synth_packed(B + P, B, 0, P - 2, synth.DIRTY)
            dch, dof = torch.from_numpy(chars).to(gpu), torch.from_numpy(offs).to(gpu)
            mask = (np.random.default_rng(B + P).random(chars.size) < 0.7).astype(np.uint8)
            dm = torch.from_numpy(mask).to(gpu)
            for d in "bf":
                for m, mdev in ((None, None), (mask, dm)):
                    exp = np.ascontiguousarray(ora.onehot_packed(chars, offs, P, d, mask=m).transpose(1, 2, 0))
                    for mb in (1, -1):
                        capi.check(lib.bsq_tuning_set(b"two_pass_slice_mb", mb))
                        got = tok.onehot_packed(dch, dof, P, d, mask=mdev, layout="bcl").cpu().numpy()
                        assert got.tobytes() == exp.tobytes(), (key, B, P, d, m is not None, mb)
    finally:
        capi.check(lib.bsq_tuning_set(b"bcl_path", 0))
        capi.check(lib.bsq_tuning_set(b"two_pass_slice_mb", 0))
