"""SURVEY.md section 8 row f-3 under a SHUFFLING sampler (missing #5 / item 6): batches of arbitrary
sequence indices -- repeats, empty sequences, any order -- rebuilt on the device from the resident FlatFile
(bsq_gather_packed_device) and encoded there.  Reference behaviour: FlatFileDataset.__getitem__ tokenises ff.access(i) per
item on the host (bioseq/loaders.py:76-104); expected values are the CPU oracle's encode of the same sequences in the
same order.  A shuffled epoch must not copy anything host -> device per batch."""
import ctypes

import numpy as np
import pytest

from bioseq_amd import synth

pytestmark = pytest.mark.gpu


def make_store(tmp_path, seed=17, n=3000, lo=0, hi=300, letters=None):
    from bioseq_amd.flatfile import FlatFile, write_flatfile
    chars, offs = synth.synth_packed(seed, n, lo, hi, letters or synth.AA)
    seqs = synth.unpack(chars, offs)
    return FlatFile(write_flatfile(seqs, str(tmp_path / "store.ff"))), seqs


def test_gather_c_abi_rebuilds_the_packed_batch(gpu, tmp_path):
    import torch
    from bioseq_amd import capi
    lib = capi.load()
    ff, seqs = make_store(tmp_path, n=2000, lo=0, hi=200, letters=synth.DIRTY)
    chars, offs = ff.to_device(gpu)
    rng = np.random.default_rng(3)
    empties = [i for i, s in enumerate(seqs) if len(s) == 0]
    assert empties
    for n in (0, 1, 5, 64, 1000, 5000):
        idx = rng.integers(0, len(seqs), size=n)
        if n >= 5:
            idx[:2] = idx[2]                    # repeats
            idx[3] = empties[0]                 # an empty sequence
            idx[4] = len(seqs) - 1              # the last one (ends at the end of the store)
        want = b"".join(bytes(seqs[i]) for i in idx)
        want_offs = np.concatenate([[0], np.cumsum([len(seqs[i]) for i in idx])]).astype(np.int64)
        d_idx = torch.from_numpy(idx.astype(np.int64)).to(gpu)
        cap = max(1, n * 200)
        out_c = torch.full((cap + 32,), 0xEE, dtype=torch.uint8, device=gpu)
        out_o = torch.empty(n + 1, dtype=torch.int64, device=gpu)
        st = torch.empty(1, dtype=torch.int64, device=gpu)
        capi.check(lib.bsq_gather_packed_device(chars.data_ptr(), offs.data_ptr(), len(seqs), d_idx.data_ptr(), n,
                                                out_c.data_ptr(), cap, out_o.data_ptr(), st.data_ptr(), None))
        assert int(st.item()) == -1
        assert (out_o.cpu().numpy() == want_offs).all()
        host = out_c.cpu().numpy()
        assert host[:len(want)].tobytes() == want
        assert (host[cap:] == 0xEE).all()                                  # nothing behind the capacity
    # bad indices and a capacity that is too small are reported in the status word, never written out of bounds
    idx = torch.tensor([3, len(seqs), 5, -1], dtype=torch.int64, device=gpu)
    out_c = torch.full((1024,), 0xEE, dtype=torch.uint8, device=gpu)
    out_o = torch.empty(5, dtype=torch.int64, device=gpu)
    st = torch.empty(1, dtype=torch.int64, device=gpu)
    capi.check(lib.bsq_gather_packed_device(chars.data_ptr(), offs.data_ptr(), len(seqs), idx.data_ptr(), 4, out_c.data_ptr(), 1024,
                                            out_o.data_ptr(), st.data_ptr(), None))
    assert int(st.item()) == 1
    o = out_o.cpu().numpy()
    assert o[2] - o[1] == 0 and o[4] - o[3] == 0                           # invalid entries contribute empty sequences
    long_ones = [i for i, s in enumerate(seqs) if len(s) > 100][:3]
    idx = torch.tensor(long_ones, dtype=torch.int64, device=gpu)
    out_c = torch.full((300,), 0xEE, dtype=torch.uint8, device=gpu)
    capi.check(lib.bsq_gather_packed_device(chars.data_ptr(), offs.data_ptr(), len(seqs), idx.data_ptr(), 3, out_c.data_ptr(), 150,
                                            out_o.data_ptr(), st.data_ptr(), None))
    assert int(st.item()) == 3 + 1                                         # output sequence 1 did not fit
    assert (out_c.cpu().numpy()[150:] == 0xEE).all()


@pytest.mark.parametrize("cnn", [False, True])
def test_shuffled_batches_equal_the_oracle_on_the_permuted_sequences(gpu, bsq, oracle, tmp_path, cnn):
    import torch
    from bioseq_amd.loaders import FlatFileDataset
    ff, seqs = make_store(tmp_path, n=1500, lo=0, hi=180)
    tok, ora = bsq.pbeos_tokenizers["PROTEIN"], oracle.OracleTokenizer("PROTEIN", 1, 1, 1)
    ds = FlatFileDataset(ff, tok, cnn=cnn, device=gpu)
    P = ds.max_seq_len

    def expect(idx):
        pick = [seqs[i] for i in idx]
        if cnn:
            return np.ascontiguousarray(ora.batch_onehot_encode(pick, padlen=P, destchar="f").transpose(1, 2, 0))
        return ora.batch_tokenize(pick, padlen=P, batch_first=True).astype(np.int64)

    rng = np.random.default_rng(11)
    for n in (1, 7, 257, 1500, 4000):                       # 4000 > len(ds): indices repeat
        idx = rng.integers(0, 1500, size=n).tolist()
        got = ds.__getitems__(idx)                           # host index list: validated here, 8 bytes per index uploaded
        assert got.is_cuda and got.dtype == (torch.float32 if cnn else torch.int64)
        assert got.cpu().numpy().tobytes() == expect(idx).tobytes()
        got = ds.__getitems__(torch.tensor(idx, device=gpu))  # device index tensor
        assert got.cpu().numpy().tobytes() == expect(idx).tobytes()
    assert ds.__getitems__([-1, 0, -1500]).cpu().numpy().tobytes() == expect([1499, 0, 0]).tobytes()   # Python indexing
    with pytest.raises(IndexError):
        ds.__getitems__([0, 1500])
    with pytest.raises(IndexError):
        ff.gather_device(torch.tensor([0, 1500], device=gpu), gpu)           # device indices are checked on the device
    # a shuffled DataLoader epoch (host sampler): every sequence once, each batch equal to the oracle on its indices
    g = torch.Generator().manual_seed(5)
    sampler = torch.utils.data.BatchSampler(torch.utils.data.RandomSampler(ds, generator=g), batch_size=256, drop_last=False)
    batches = list(sampler)
    dl = torch.utils.data.DataLoader(ds, batch_sampler=batches, collate_fn=lambda x: x)
    seen = []
    for idx, got in zip(batches, dl):
        assert got.cpu().numpy().tobytes() == expect(idx).tobytes()
        seen += idx
    assert sorted(seen) == list(range(1500))


def test_a_shuffled_epoch_copies_nothing_host_to_device(gpu, bsq, oracle, tmp_path, monkeypatch):
    """The device-side sampler (`FlatFileDataset.batches`): permutation drawn on the device, index tensors never leave HBM,
    batches gathered + augmented + encoded there.  Counters: the library's own host -> device byte count
    (bsq_host_upload_bytes) and every torch host -> device path (Tensor.to / .cuda / torch.tensor(device=) / from_numpy)."""
    import torch
    from bioseq_amd import capi
    from bioseq_amd.loaders import AugmentedSeqDataset, FlatFileDataset
    lib = capi.load()
    ff, seqs = make_store(tmp_path, n=2048, lo=1, hi=120)
    tok, ora = bsq.Tokenizer("SEB8", 1, 1, 1), oracle.OracleTokenizer("SEB8", 1, 1, 1)
    ds = FlatFileDataset(ff, tok, device=gpu)
    aug = AugmentedSeqDataset(ff, tok, device=gpu)
    ds.get_batch(0, 4), aug.get_batch(0, 4)                  # upload the store, build the tables: untimed set-up
    torch.cuda.synchronize()
    uploads = []
    real_to, real_cuda = torch.Tensor.to, torch.Tensor.cuda

    def to(self, *a, **k):
        out = real_to(self, *a, **k)
        if not self.is_cuda and out.is_cuda:
            uploads.append(("to", self.numel() * self.element_size()))
        return out

    def cuda(self, *a, **k):
        if not self.is_cuda:
            uploads.append(("cuda", self.numel() * self.element_size()))
        return real_cuda(self, *a, **k)
    readbacks = []
    real_item = torch.Tensor.item

    def item(self):
        if self.is_cuda:
            readbacks.append("item")  # (a synchronising read-back: round 3 paid one per batch for the gather's status word)
        return real_item(self)
    monkeypatch.setattr(torch.Tensor, "to", to)
    monkeypatch.setattr(torch.Tensor, "cuda", cuda)
    monkeypatch.setattr(torch.Tensor, "item", item)
    before = lib.bsq_host_upload_bytes()
    g = torch.Generator(device=gpu).manual_seed(99)
    total, order = 0, []
    for batch in ds.batches(200, shuffle=True, generator=g):
        assert batch.is_cuda and batch.dtype == torch.int64
        total += batch.shape[0]
        order.append(batch)
    for batch in aug.batches(200, shuffle=True):
        assert batch.is_cuda
    torch.cuda.synchronize()
    assert lib.bsq_host_upload_bytes() == before, "the library copied host -> device during the epoch"
    assert uploads == [], uploads
    assert readbacks == [], "the epoch's own permutation needs no check: nothing is read back, no batch synchronises"
    monkeypatch.undo()
    assert total == 2048
    # the epoch is a permutation of the oracle's rows (same generator -> same order)
    g = torch.Generator(device=gpu).manual_seed(99)
    perm = torch.randperm(2048, device=gpu, generator=g).cpu().numpy()
    exp = ora.batch_tokenize([seqs[i] for i in perm], padlen=ds.max_seq_len, batch_first=True).astype(np.int64)
    assert torch.cat(order).cpu().numpy().tobytes() == exp.tobytes()


def test_a_loader_step_as_one_hip_graph(gpu, bsq, oracle, tmp_path):
    """Gather + augmentation + encode of an index batch are plain stream-ordered launches: captured ONCE into a HIP graph and
    replayed on new index lists (a launch-bound loader step becomes one graph launch).  Expected: the oracle on the gathered
    sequences (augmentation off) / exactly one residue changed in about half of them (on)."""
    import torch
    from bioseq_amd import blosum
    ff, seqs = make_store(tmp_path, n=4000, lo=1, hi=200)
    tok, ora = bsq.Tokenizer("AMINO20", 1, 1, 1), oracle.OracleTokenizer("AMINO20", 1, 1, 1)
    P, nb = 202, 512
    ff.to_device(gpu)
    idx = torch.zeros(nb, dtype=torch.int64, device=gpu)
    rng = np.random.default_rng(8)

    def step(augment):
        chars, offs = ff.gather_device(idx, gpu, validate=False)      # validate=False: no synchronising status read inside a capture
        if augment:
            blosum.augment_packed(chars, offs, 1, 0.5, 1234)
        return tok.tokenize_packed(chars, offs, P, "q", True, validate=False)

    for augment in (False, True):
        step(augment)                                                 # warm-up outside the capture (tables, scratch)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = step(augment)
        for rep in range(3):
            pick = rng.integers(0, 4000, size=nb)
            idx.copy_(torch.from_numpy(pick))
            out.zero_()
            g.replay()
            torch.cuda.synchronize()
            exp = ora.batch_tokenize([seqs[i] for i in pick], padlen=P, batch_first=True).astype(np.int64)
            got = out.cpu().numpy()
            if not augment:
                assert got.tobytes() == exp.tobytes()
            else:
                ndiff = (got != exp).sum(axis=1)
                assert set(np.unique(ndiff)) <= {0, 1} and 0.35 * nb < (ndiff == 1).sum() < 0.65 * nb


@pytest.mark.parametrize("forced_general", [0, 1], ids=["one-launch", "three-launches"])
def test_gather_mid_and_large_lists(gpu, tmp_path, forced_general):
    """Lists beyond the 4096 indices of k_gather_small: k_gather_mid (one launch, up to 65 536) and the three-launch path behind it
    (and, knob gather_small = 1, for every size) -- offsets and characters against numpy, repeats / empty sequences / bad indices /
    a capacity that is too small included."""
    import torch
    from bioseq_amd import capi
    lib = capi.load()
    ff, seqs = make_store(tmp_path, n=3000, lo=0, hi=90, letters=synth.DIRTY)
    chars, offs = ff.to_device(gpu)
    lens = np.array([len(s) for s in seqs], dtype=np.int64)
    store = np.frombuffer(b"".join(bytes(s) for s in seqs), dtype=np.uint8)
    starts = np.concatenate([[0], np.cumsum(lens)])
    rng = np.random.default_rng(77)
    capi.check(lib.bsq_tuning_set(b"gather_small", forced_general))
    try:
        for n in (4097, 8191, 16384, 20001, 65536, 65537, 70003):
            idx = rng.integers(0, len(seqs), size=n).astype(np.int64)
            want_offs = np.concatenate([[0], np.cumsum(lens[idx])]).astype(np.int64)
            pos = np.repeat(starts[idx] - want_offs[:-1], lens[idx]) + np.arange(want_offs[-1])
            want = store[pos]
            d_idx = torch.from_numpy(idx).to(gpu)
            cap = int(want_offs[-1])
            out_c = torch.full((cap + 64,), 0xEE, dtype=torch.uint8, device=gpu)
            out_o = torch.full((n + 1,), -5, dtype=torch.int64, device=gpu)
            st = torch.empty(1, dtype=torch.int64, device=gpu)
            capi.check(lib.bsq_gather_packed_device(chars.data_ptr(), offs.data_ptr(), len(seqs), d_idx.data_ptr(), n, out_c.data_ptr(), cap,
                                                    out_o.data_ptr(), st.data_ptr(), None))
            assert int(st.item()) == -1
            assert (out_o.cpu().numpy() == want_offs).all(), n
            host = out_c.cpu().numpy()
            assert (host[:cap] == want).all(), n
            assert (host[cap:] == 0xEE).all()
            # without a status word (a caller that vouches for its indices): the same batch
            out_c2 = torch.full((cap + 64,), 0xEE, dtype=torch.uint8, device=gpu)
            capi.check(lib.bsq_gather_packed_device(chars.data_ptr(), offs.data_ptr(), len(seqs), d_idx.data_ptr(), n, out_c2.data_ptr(), cap,
                                                    out_o.data_ptr(), None, None))
            assert torch.equal(out_c, out_c2)
        # bad indices contribute empty sequences and are reported (the first one); a short capacity is cut, reported, never overrun
        n = 9000
        idx = rng.integers(0, len(seqs), size=n).astype(np.int64)
        idx[[17, 4500, 8999]] = [len(seqs), -1, 1 << 40]
        d_idx = torch.from_numpy(idx).to(gpu)
        ok = (idx >= 0) & (idx < len(seqs))
        ln = np.where(ok, lens[np.where(ok, idx, 0)], 0)
        want_offs = np.concatenate([[0], np.cumsum(ln)]).astype(np.int64)
        out_c = torch.full((int(want_offs[-1]) + 64,), 0xEE, dtype=torch.uint8, device=gpu)
        out_o = torch.empty(n + 1, dtype=torch.int64, device=gpu)
        st = torch.empty(1, dtype=torch.int64, device=gpu)
        capi.check(lib.bsq_gather_packed_device(chars.data_ptr(), offs.data_ptr(), len(seqs), d_idx.data_ptr(), n, out_c.data_ptr(), int(want_offs[-1]),
                                                out_o.data_ptr(), st.data_ptr(), None))
        assert int(st.item()) == 17 and (out_o.cpu().numpy() == want_offs).all()
        idx = rng.integers(0, len(seqs), size=n).astype(np.int64)
        d_idx = torch.from_numpy(idx).to(gpu)
        want_offs = np.concatenate([[0], np.cumsum(lens[idx])]).astype(np.int64)
        cap = int(want_offs[6000]) + 3
        out_c = torch.full((int(want_offs[-1]) + 64,), 0xEE, dtype=torch.uint8, device=gpu)
        capi.check(lib.bsq_gather_packed_device(chars.data_ptr(), offs.data_ptr(), len(seqs), d_idx.data_ptr(), n, out_c.data_ptr(), cap,
                                                out_o.data_ptr(), st.data_ptr(), None))
        first_cut = int(np.argmax(want_offs[1:] > cap))
        assert int(st.item()) == n + first_cut
        host = out_c.cpu().numpy()
        assert (host[cap:] == 0xEE).all()
        pos = np.repeat(starts[idx] - want_offs[:-1], lens[idx]) + np.arange(want_offs[-1])
        assert (host[:cap] == store[pos][:cap]).all()
    finally:
        capi.check(lib.bsq_tuning_set(b"gather_small", 0))
