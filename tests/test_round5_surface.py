"""GPU: the round-5 additions to the drop-in surface against the oracle -- `FlatFileDataset.fetch` with the reference's return
shapes (bioseq/loaders.py:65-84), a store opened with a too-small maxseqlen (validated, raises -- the reference aborts,
tokenize.h:359-362), and a list batch whose characters outgrow the staging area's sampled estimate (falls back, same result)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_fetch_has_the_reference_shapes(gpu, bsq, oracle, tmp_path):
    import torch
    from bioseq_amd import synth
    from bioseq_amd.flatfile import FlatFile, write_flatfile
    from bioseq_amd.loaders import FlatFileDataset
    chars, offs = synth.synth_packed(77, 300, 1, 90, synth.AA)
    seqs = synth.unpack(chars, offs)
    ff = FlatFile(write_flatfile(seqs, str(tmp_path / "f.ff")))
    tok, ora = bsq.pbeos_tokenizers["PROTEIN"], oracle.OracleTokenizer("PROTEIN", 1, 1, 1)
    P = ff.maxseqlen + 2
    exp_oh = np.ascontiguousarray(ora.batch_onehot_encode(seqs, padlen=P, destchar="f").transpose(1, 2, 0))   # 'l b c -> b c l'
    exp_tok = ora.batch_tokenize(seqs, padlen=P, batch_first=True)
    cnn = FlatFileDataset(ff, tok, cnn=True, device=gpu)
    r = cnn.fetch(slice(10, 42))                                       # list index: (B, C, L) float32 on the device
    assert r.is_cuda and r.dtype == torch.float32 and tuple(r.shape) == (32, tok.alphabet_size(), P)
    assert r.cpu().numpy().tobytes() == exp_oh[10:42].tobytes()
    r, items = cnn.fetch(np.array([5, 299, 17]), return_items=True)    # index array + the items themselves
    assert r.cpu().numpy().tobytes() == exp_oh[[5, 299, 17]].tobytes()
    assert [bytes(x) for x in items] == [seqs[5], seqs[299], seqs[17]] and isinstance(items[0], bytearray)
    r = cnn.fetch(slice(0, 300, 7))
    assert r.cpu().numpy().tobytes() == exp_oh[0:300:7].tobytes()
    one, item = cnn.fetch(9, return_items=True)                        # single int: the single-sequence one-hot, not rearranged
    exp_one = ora.onehot_encode(seqs[9], padlen=P, destchar="f") if hasattr(ora, "onehot_encode") else None
    L = len(seqs[9])
    assert one.dtype == torch.float32 and tuple(one.shape) == (max(L, P) + 2, tok.alphabet_size()) and bytes(item) == seqs[9]
    if exp_one is not None:
        assert one.cpu().numpy().tobytes() == np.asarray(exp_one, dtype=np.float32).tobytes()
    assert one.cpu().numpy()[:L + 2].tobytes() == exp_oh[9].T[:L + 2].tobytes()   # BOS, residues, EOS rows = the batch form's
    plain = FlatFileDataset(ff, tok, device=gpu)
    row = plain.fetch(9, return_items=True)                            # cnn=False: the token row, return_items ignored
    assert row.dtype == torch.long and tuple(row.shape) == (P,) and (row.cpu().numpy() == exp_tok[9]).all()
    aug = FlatFileDataset(ff, tok, cnn=True, augment=1, augment_frac=1.0, device=gpu)
    r, items = aug.fetch(slice(0, 64), return_items=True)              # the returned items are the MUTATED sequences
    diff = [sum(a != b for a, b in zip(bytes(x), s)) for x, s in zip(items, seqs[:64])]
    assert set(diff) == {1}
    exp_mut = np.ascontiguousarray(ora.batch_onehot_encode([bytes(x) for x in items], padlen=P, destchar="f").transpose(1, 2, 0))
    assert r.cpu().numpy().tobytes() == exp_mut.tobytes()
    assert [bytes(x) for x in ff.access(0, 64)] == seqs[:64]          # the store itself is untouched


def test_a_store_opened_with_a_too_small_maxseqlen_is_validated(gpu, bsq, tmp_path):
    from bioseq_amd.flatfile import FlatFile, write_flatfile
    from bioseq_amd.loaders import FlatFileDataset
    seqs = [b"ACDEFGHIKL" * 3, b"ACD", b"MKV" * 20, b"WW"]                 # the third one is 60 long
    path = write_flatfile(seqs, str(tmp_path / "s.ff"))
    tok = bsq.pbeos_tokenizers["PROTEIN"]
    for kw in ({}, {"cnn": True}, {"augment": 1, "augment_frac": 1.0}, {"augment": 1, "augment_frac": 1.0, "token_dtype": "b"}):
        ds = FlatFileDataset(FlatFile(path, maxseqlen=40), tok, device=gpu, **kw)
        assert ds.max_seq_len == 42 and not ds._trusted_lengths
        assert tuple(ds.get_batch(0, 2).shape)[0] == 2                      # batches of legal sequences still encode
        with pytest.raises((ValueError, RuntimeError), match="seq len \\+ bos \\+ eos > padlen"):
            ds.get_batch(0, 4)                                              # never a silently truncated sequence
        with pytest.raises((ValueError, RuntimeError), match="seq len \\+ bos \\+ eos > padlen"):
            ds.__getitems__([3, 2])
    assert FlatFileDataset(FlatFile(path), tok, device=gpu)._trusted_lengths


@pytest.mark.parametrize("to_device", [True, False])
def test_a_list_that_outgrows_the_staging_estimate_falls_back(gpu, bsq, oracle, to_device):
    """The staged paths size their pinned / device areas from <= 512 sampled items (sizing by n * padlen pinned
    GiBs for short sequences under a generous padlen); long items the sample never sees overflow the estimate, and the call must
    fall back to the whole-batch path with the oracle's result."""
    n, P = 16384 + 64, 4100
    items = [b"ACGT" * 2] * n
    long_ = b"ACGTTGCA" * 500
    for i in range(1, n, 32):   # the sample reads items k * n / 512: never index = 1 mod 32 ... (n / 512 = 32.125: a few are hit)
        items[i] = long_
    tok, ora = bsq.Tokenizer("DNA", True, True, True), oracle.OracleTokenizer("DNA", True, True, True)
    exp = ora.batch_tokenize(items, padlen=P, batch_first=True)
    got = tok.batch_tokenize(items, padlen=P, batch_first=True, device=gpu if to_device else None)
    got = got.cpu().numpy() if to_device else got
    assert got.tobytes() == exp.tobytes()
    # and a generous padlen over short sequences no longer pins n * padlen bytes: the call works and is right
    short = [b"ACGTN"[: 1 + (i % 5)] for i in range(20000)]
    P2 = 60000 if to_device else 6000   # (numpy results are staged up to 256 MB)
    exp = ora.batch_tokenize(short, padlen=P2, batch_first=True)
    got = tok.batch_tokenize(short, padlen=P2, batch_first=True, device=gpu if to_device else None)
    got = got.cpu().numpy() if to_device else got
    assert got.tobytes() == exp.tobytes()


def test_release_staging_between_calls(gpu, bsq, oracle):
    """`cbioseq.release_staging()` gives the pinned ring and the device staging areas back (a long-lived process that is done with a phase of
    list -> device calls); the next call of every host form builds them again and is exact -- small and staged-in-pieces batches, numpy and
    device results, also while another thread is encoding"""
    import threading
    from bioseq_amd import cbioseq, synth
    tok, ora = bsq.Tokenizer("PROTEIN", 1, 1, 1), oracle.OracleTokenizer("PROTEIN", 1, 1, 1)
    P = 96
    small = synth.unpack(*synth.synth_packed(41, 300, 0, P - 2, synth.AA))
    big = synth.unpack(*synth.synth_packed(42, 40000, 0, P - 2, synth.AA))     # >= 16384 items: the staged path in pieces
    want = {id(small): (ora.batch_tokenize(small, padlen=P, batch_first=True), ora.batch_onehot_encode(small, padlen=P, destchar="f")),
            id(big): (ora.batch_tokenize(big, padlen=P, batch_first=True), ora.batch_onehot_encode(big, padlen=P, destchar="f"))}

    def check(seqs):
        wt, wo = want[id(seqs)]
        assert tok.batch_tokenize(seqs, padlen=P, batch_first=True).tobytes() == wt.tobytes()
        assert tok.batch_tokenize(seqs, padlen=P, batch_first=True, device=gpu).cpu().numpy().tobytes() == wt.tobytes()
        assert tok.batch_onehot_encode(seqs, padlen=P, destchar="f", device=gpu).cpu().numpy().tobytes() == wo.tobytes()

    for _ in range(2):
        check(small), check(big)
        cbioseq.release_staging()
        cbioseq.release_staging()
    errors = []

    def worker():
        try:
            for _ in range(3):
                check(big)
        except Exception as ex:  # noqa: BLE001
            errors.append(ex)
    t = threading.Thread(target=worker)
    t.start()
    for _ in range(20):
        cbioseq.release_staging()   # waits for a staged batch in flight, never tears its buffers away
    t.join()
    assert not errors, errors
    check(small)
