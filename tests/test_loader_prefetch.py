"""`FlatFileDataset.batches(prefetch=k)` (round 6): the next k batches are gathered + augmented + encoded on two side streams while
the consumer holds the current one (reference: the training loop over `DataLoader(FlatFileDataset)`, bioseq/loaders.py:76-104, where
consecutive batches are independent).  The batches must be the tensors of the in-order epoch bit for bit -- which the oracle
tests of test_index_batches.py / test_bcl_and_loaders.py pin --, here additionally against the oracle directly, with a consumer that
works on its own stream, for every dataset kind."""
import numpy as np
import pytest

from bioseq_amd import synth

pytestmark = pytest.mark.gpu


def make_store(tmp_path, seed=23, n=3000, lo=0, hi=200):
    from bioseq_amd.flatfile import FlatFile, write_flatfile
    chars, offs = synth.synth_packed(seed, n, lo, hi, synth.AA)
    seqs = synth.unpack(chars, offs)
    return FlatFile(write_flatfile(seqs, str(tmp_path / "store.ff"))), seqs


@pytest.mark.parametrize("kind", ["tokens", "tokens8", "cnn", "augment", "augment8", "augment_cnn"])
@pytest.mark.parametrize("shuffle", [True, False], ids=["shuffled", "in-order"])
def test_prefetched_epoch_equals_the_in_order_epoch(gpu, bsq, tmp_path, kind, shuffle):
    import torch
    from bioseq_amd.loaders import AugmentedSeqDataset, FlatFileDataset
    ff, _ = make_store(tmp_path)
    tok = bsq.Tokenizer("SEB8", 1, 1, 1)
    kw = {"cnn": "cnn" in kind, "token_dtype": "b" if kind.endswith("8") else "q", "device": gpu}

    def epoch(prefetch, consume=False):
        ds = (AugmentedSeqDataset(ff, tok, seed=5, **kw) if kind.startswith("augment") else FlatFileDataset(ff, tok, **kw))
        g = torch.Generator(device=gpu).manual_seed(1234)
        out = []
        for batch in ds.batches(256, shuffle=shuffle, generator=g, prefetch=prefetch):
            if consume:  # the consumer works on the batch on ITS stream right away (the event hand-off must order it)
                out.append((batch.clone(), batch.to(torch.float64).sum()))
            else:
                out.append((batch.clone(), None))
        torch.cuda.synchronize()
        return out

    base = epoch(0)
    assert len(base) == 12 and base[-1][0].shape[0] == 3000 - 11 * 256
    for k in (1, 2, 5, 50):
        got = epoch(k, consume=True)
        assert len(got) == len(base)
        for (a, _), (b, s) in zip(base, got):
            assert a.shape == b.shape and torch.equal(a, b), (kind, k)
            assert float(s) == float(a.to(torch.float64).sum())


def test_prefetched_epoch_vs_oracle_with_a_side_stream_consumer(gpu, bsq, oracle, tmp_path):
    """The consumer runs on a stream of its own (not the default one) and reads every batch there without any synchronisation of its
    own; drop_last; the dataset-level default `prefetch=`; expected values straight from the oracle."""
    import torch
    from bioseq_amd.loaders import FlatFileDataset
    ff, seqs = make_store(tmp_path, n=1000, hi=150)
    tok, ora = bsq.Tokenizer("PROTEIN", 1, 1, 1), oracle.OracleTokenizer("PROTEIN", 1, 1, 1)
    ds = FlatFileDataset(ff, tok, device=gpu, prefetch=2)
    P = ds.max_seq_len
    g = torch.Generator(device=gpu).manual_seed(7)
    perm = torch.randperm(1000, device=gpu, generator=torch.Generator(device=gpu).manual_seed(7)).cpu().tolist()
    mine = torch.cuda.Stream(device=gpu)
    sums, kept = [], []
    with torch.cuda.stream(mine):
        for batch in ds.batches(128, shuffle=True, drop_last=True, generator=g):
            sums.append(batch.sum())
            kept.append(batch)
    mine.synchronize()
    assert len(kept) == 7
    for k, batch in enumerate(kept):
        want = ora.batch_tokenize([seqs[i] for i in perm[k * 128:(k + 1) * 128]], padlen=P, batch_first=True).astype(np.int64)
        assert batch.cpu().numpy().tobytes() == want.tobytes(), k
        assert int(sums[k]) == int(want.sum())


@pytest.mark.parametrize("kind", ["tokens", "tokens8", "cnn"])
@pytest.mark.parametrize("shuffle", [True, False], ids=["shuffled", "in-order"])
def test_grouped_epoch_equals_the_plain_epoch(gpu, bsq, tmp_path, kind, shuffle):
    """`batches(group=G)`: G consecutive batches gathered + encoded as one super-batch and handed out as its row blocks -- bit for
    bit the batches of group = 1 (without augmentation), with and without prefetching, drop_last and a ragged tail included."""
    import torch
    from bioseq_amd.loaders import FlatFileDataset
    ff, _ = make_store(tmp_path, n=3100)
    tok = bsq.Tokenizer("SEB8", 1, 1, 1)
    kw = {"cnn": kind == "cnn", "token_dtype": "b" if kind.endswith("8") else "q", "device": gpu}

    def epoch(**opts):
        ds = FlatFileDataset(ff, tok, **kw)
        g = torch.Generator(device=gpu).manual_seed(99)
        out = [b.clone() for b in ds.batches(200, shuffle=shuffle, generator=g, **opts)]
        torch.cuda.synchronize()
        return out

    for drop_last in (False, True):
        base = epoch(drop_last=drop_last)
        assert len(base) == (15 if drop_last else 16) and base[0].shape[0] == 200 and base[-1].shape[0] == (200 if drop_last else 100)
        for G, pf in ((2, 0), (4, 0), (7, 0), (4, 2), (16, 1), (64, 3)):
            got = epoch(drop_last=drop_last, group=G, prefetch=pf)
            assert [tuple(x.shape) for x in got] == [tuple(x.shape) for x in base], (G, pf, drop_last)
            for a, b in zip(base, got):
                assert torch.equal(a, b), (G, pf, drop_last)


def test_grouped_augmented_epoch_is_one_draw_per_super_batch(gpu, bsq, tmp_path):
    """With `augment`, a super-batch is mutated by ONE call (the seed of its first batch, sequences numbered across the super-batch):
    its batches are the row blocks of the augmented encode of the super-batch's sequences -- checked against that very call."""
    import torch
    from bioseq_amd import blosum
    from bioseq_amd.loaders import AugmentedSeqDataset
    ff, _ = make_store(tmp_path, n=2000)
    tok = bsq.Tokenizer("SEB8", 1, 1, 1)
    ds = AugmentedSeqDataset(ff, tok, device=gpu, token_dtype="b", seed=3)
    got = [b.clone() for b in ds.batches(250, shuffle=False, group=4)]
    assert [b.shape[0] for b in got] == [250] * 8
    ref = AugmentedSeqDataset(ff, tok, device=gpu, token_dtype="b", seed=3)
    for k in range(2):
        big = ref.get_batch(1000 * k, 1000 * (k + 1))  # the same calls in the same order: seeds 4, 5
        for j in range(4):
            assert torch.equal(got[4 * k + j], big[250 * j:250 * (j + 1)])
    changed = sum(int((a != b).any(dim=1).sum()) for a, b in zip(got, [x.clone() for x in AugmentedSeqDataset(ff, tok, device=gpu, token_dtype="b", augment=0).batches(250, shuffle=False)]))
    assert 0.1 * 2000 < changed < 0.6 * 2000   # (half of the sequences mutate; a substitution inside one SEB8 class leaves the row as it was)
    blosum.check_fused(synchronize=True)
