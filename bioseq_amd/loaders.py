"""Dataset layer over FlatFile + Tokenizer -- the surface of /root/reference/bioseq/loaders.py, with the
batch as the unit of work.

The reference's `FlatFileDataset.__getitem__` tokenises ONE sequence per call on the host
(`batch_tokenize([seq], ...)`, loaders.py:84,103) and the DataLoader stacks the results.  Here a batch is
encoded by ONE launch from the FlatFile's packed bytes (`get_batch`, and `__getitems__` for
torch >= 2 DataLoader batched fetching), optionally after BLOSUM62 augmentation on the device, and the
conv-net layout 'length batch emb -> batch emb length' + .float() (loaders.py:74) is written directly
by the kernel (`layout="bcl"`, float32) instead of being produced by a permute + cast.
"""
from __future__ import annotations

import numpy as np
import torch

from . import blosum
from .flatfile import FlatFile


def FF2NP(x, tokenizer, destfile, *, batch_size=8192):
    """Token matrix of a whole FlatFile as a uint8 memmap at `destfile`: (nseqs, maxseqlen + bos + eos), batch-first
    (the reference's helper of the same name, loaders.py:11-26) -- `batch_size` sequences per device encode."""
    if not isinstance(x, FlatFile):
        raise TypeError("FF2NP expects a FlatFile")
    n = x.nseqs()
    width = x.maxseqlen + int(tokenizer.includes_bos()) + int(tokenizer.includes_eos())
    tokens = np.memmap(destfile, dtype=np.uint8, mode="w+", shape=(n, width))
    first = 0
    while first < n:
        last = min(n, first + batch_size)
        rows = x.batch_tokenize(tokenizer, first, last, padlen=width, destchar="B", batch_first=True)
        tokens[first:last] = rows.view(np.uint8)
        first = last
    return tokens, destfile


class FlatFileDataset(torch.utils.data.Dataset):
    """FlatFile + Tokenizer dataset (loaders.py:29-115).

    cnn=False: items are token rows (max_seq_len,) int64; cnn=True: one-hot (C, max_seq_len) float32.
    `augment` BLOSUM62 point mutations are applied to a sequence with probability `augment_frac`
    (>= 1: always) before encoding.  `get_batch(start, stop)` / `__getitems__(indices)` return the
    stacked batch -- (B, P) int64 or (B, C, P) float32 on `device` -- from a single encode.
    """

    def __init__(self, ff, tokenizer, *, augment=0, augment_frac=0.5, cnn=False, device=None, maskfrac=0.15, seed=13, token_dtype="q", prefetch=0):
        super().__init__()
        if not isinstance(ff, FlatFile):
            raise TypeError("FlatFileDataset expects a FlatFile")
        self.device = torch.device("cuda" if device is None else device)
        self.ff, self.tokenizer = ff, tokenizer
        # FlatFile(path, maxseqlen=N) takes the caller's word for N; when a stored sequence is longer than that, max_seq_len is too
        # short for it: the reference aborts on such a sequence (tokenize.h:359-362), so batches are then validated (and raise) instead
        # of being trusted (unvalidated kernels would clamp the sequence silently)
        self._trusted_lengths = ff.maxseqlen >= getattr(ff, "_longest", ff.maxseqlen + 1)
        # attribute names as in the reference's class (loaders.py:36-47)
        self.max_seq_len = self.maxseqlen = ff.maxseqlen + int(tokenizer.includes_bos()) + int(tokenizer.includes_eos())
        self.augment, self.augment_frac = augment, augment_frac
        self.cnn, self.maskfrac = cnn, maskfrac
        self._seed = int(seed)
        self._calls = 0
        # element type of the token rows: 'q' = int64 as the reference's loader hands them to nn.Embedding (loaders.py:84-86);
        # 'b' = int8 -- an eighth of the bytes, and with `augment` set the whole batch step is ONE launch (bsq_augment_tokenize_device)
        self.token_dtype = token_dtype
        # `batches()`: how many batches are encoded ahead of the consumer on the two side streams (0: in order on the current stream)
        self.prefetch = int(prefetch)
        self._side = None

    def __len__(self):
        return self.ff.nseqs()

    def _packed_device(self, start, stop, indices=None, trusted=False):
        """The batch's own packed copy on the device (never the resident store when it is going to be mutated).
        trusted: `indices` is this dataset's own permutation (in range, no repeats) -- nothing to check, nothing read back."""
        if indices is None:
            chars, offs = self.ff.packed_device(start, stop, self.device)
            if self.augment:
                chars = chars.clone()  # never mutate the resident store
        else:  # arbitrary index set (a shuffling sampler): rebuilt on the device from the resident store, no host gather
            chars, offs = self.ff.gather_device(indices, self.device, validate=not trusted, distinct=trusted)
        return chars, offs

    def _encode(self, chars, offs):
        """augment_seq, then encode (bioseq/loaders.py:83-84, :102-103) on the batch's own copy.  Token rows go through the
        one-call entry `blosum.augment_tokenize_packed` (one launch for int8 rows; the entry runs the two launches for the
        other types); the one-hot form augments, then encodes."""
        trusted = self._trusted_lengths
        if self.augment:
            self._calls += 1
            seed = self._seed + self._calls
            if not self.cnn and trusted:
                return blosum.augment_tokenize_packed(self.tokenizer, chars, offs, self.max_seq_len, self.token_dtype, True,
                                                      chain_len=self.augment, augment_frac=self.augment_frac, seed=seed)
            blosum.augment_packed(chars, offs, self.augment, self.augment_frac, seed)
        # validate=False only when no stored sequence is longer than ff.maxseqlen (= max_seq_len - bos - eos): every batch cut or
        # gathered from the store then has well-formed offsets and legal lengths, and the device-side check would only add a
        # synchronising read-back (~30 us) per batch.  A store opened with a smaller maxseqlen is validated and raises.
        if self.cnn:
            return self.tokenizer.onehot_packed(chars, offs, self.max_seq_len, "f", layout="bcl", validate=not trusted)
        # int64 rows written by the kernel itself ('q'), not int8 + a .to(torch.long) pass over the matrix
        return self.tokenizer.tokenize_packed(chars, offs, self.max_seq_len, self.token_dtype, True, validate=not trusted)

    def get_batch(self, start, stop):
        """Sequences [start, stop) as one encoded batch on the device."""
        return self._encode(*self._packed_device(start, stop))

    def _index(self, i):
        """Python indexing: -len <= i < len, else IndexError (the reference's ff.access raises out_of_range; a silent
        wrap would also make `for x in ds` -- the legacy iteration protocol -- run forever)."""
        n = len(self)
        i = int(i)
        if not -n <= i < n:
            raise IndexError("index %d is out of range for %d sequences" % (i, n))
        return i + n if i < 0 else i

    def __getitems__(self, indices):
        """One encoded batch ON THE DEVICE for the whole index list -- a single stacked tensor, not a list of samples:
        use `DataLoader(ds, batch_size=..., collate_fn=lambda batch: batch, num_workers=0)` (the default collate would
        re-stack it with a copy, and device tensors cannot cross worker processes).  A list of host integers costs one
        upload of 8 bytes per index; an int64 tensor on the device (see `batches`) costs none."""
        if isinstance(indices, torch.Tensor) and indices.is_cuda:
            return self._encode(*self._packed_device(0, 0, indices))
        idx = [self._index(i) for i in indices]
        if idx and idx == list(range(idx[0], idx[0] + len(idx))):
            return self.get_batch(idx[0], idx[-1] + 1)
        return self._encode(*self._packed_device(0, 0, idx))

    def batches(self, batch_size, shuffle=True, drop_last=False, generator=None, prefetch=None, group=1):
        """One epoch of encoded batches with the sampler ON THE DEVICE: a `torch.randperm` drawn on `self.device` (or the
        identity), cut into index tensors that never leave HBM.  Nothing is copied host -> device per batch: the store is
        resident (FlatFile.to_device), the indices are device tensors, the batch is gathered, augmented and encoded there.
        Yields (B, P) int64 tokens or (B, C, P) float32 one-hots, like `__getitems__`.

        Consecutive batches are independent, and this loop is the one place that knows it (round 6):

        group = G > 1   G consecutive batches are gathered and encoded as ONE super-batch (one gather launch, one encode launch) and
                        handed out as its row blocks -- both layouts are batch-first, so batch k is rows [k * batch_size, (k + 1) *
                        batch_size) of the super-batch, a view.  A batch step of <= 4096 sequences is bound by the host's ~20 us of
                        launches, not by the GPU: G = 4 brings the epoch of 4096-sequence batches from 23 to ~8 us per batch.  The
                        batches are bit for bit those of G = 1, except with `augment`: the mutations are then one draw over the
                        super-batch (seeded like its first batch) instead of G draws -- the same law, other random numbers.
        prefetch = k > 0 (None: the dataset's `prefetch` attribute)  the gather + augmentation + encode of the next k (super-)batches
                        are issued on two SIDE STREAMS that take turns while the consumer still holds the current one; a batch is
                        handed over with an event the consumer's current stream waits for (no host synchronisation) and
                        `record_stream`.  Same tensors bit for bit as prefetch = 0.  The hand-off costs ~14 us of host time per
                        (super-)batch: it pays when the encode is long enough to hide under the consumer's own kernels (large
                        batches, one-hot outputs), and LOSES in a host-bound loop of small batches (profiles/r06/loader_prefetch.txt)
                        -- hence off by default."""
        n = len(self)
        order = (torch.randperm(n, device=self.device, generator=generator) if shuffle
                 else torch.arange(n, device=self.device))
        fused = bool(self.augment) and not self.cnn and str(self.token_dtype)[:1].lower() == "b"  # int8 rows take the one-launch entry, whose in-kernel wait can (in theory) expire
        depth = int(self.prefetch if prefetch is None else prefetch)
        batch_size = int(batch_size)
        if batch_size <= 0:
            raise ValueError("batch_size must be positive")
        span = batch_size * max(1, int(group))
        n_eff = n - n % batch_size if drop_last else n
        firsts = list(range(0, n_eff, span))

        def encode(first):
            stop = min(n_eff, first + span)
            if shuffle:
                return self._encode(*self._packed_device(0, 0, order[first:stop], trusted=True))
            return self.get_batch(first, stop)

        def hand_out(big):
            if big.shape[0] <= batch_size:
                yield big
            else:
                for r in range(0, big.shape[0], batch_size):
                    yield big[r:r + batch_size]

        try:
            if depth <= 0:
                for first in firsts:
                    big = encode(first)
                    if fused:
                        blosum.check_fused()  # host memory only: the launches that have completed so far (a poisoned batch raises here or below)
                    yield from hand_out(big)
                return
            import collections
            with torch.cuda.device(self.device):
                self.ff.to_device(self.device)  # (the store goes up once, before any side stream reads it)
                consumer = torch.cuda.current_stream()
                if self._side is None:
                    self._side = (torch.cuda.Stream(), torch.cuda.Stream())
                for side in self._side:  # whatever produced the store and `order` on the consumer's stream comes first
                    side.wait_stream(consumer)
                order.record_stream(self._side[0]), order.record_stream(self._side[1])
                queue = collections.deque()
                it = iter(enumerate(firsts))
                events = [torch.cuda.Event() for _ in range(depth + 1)]  # reused in turn: at most depth + 1 batches are in flight
                set_stream = torch.cuda.set_stream  # (cheaper than entering a `torch.cuda.stream` context per batch)

                def issue():
                    k, first = next(it, (None, None))
                    if first is None:
                        return
                    side = self._side[k & 1]
                    back = torch.cuda.current_stream()
                    set_stream(side)
                    try:
                        big = encode(first)
                        ready = events[k % (depth + 1)]
                        ready.record(side)
                    finally:
                        set_stream(back)
                    queue.append((big, ready))

                for _ in range(depth):
                    issue()
                while queue:
                    big, ready = queue.popleft()
                    issue()  # the next one goes out BEFORE this one is handed over: it runs under whatever the consumer does with it
                    now = torch.cuda.current_stream()
                    now.wait_event(ready)
                    big.record_stream(now)
                    if fused:
                        blosum.check_fused()
                    yield from hand_out(big)
        finally:
            if fused:  # the epoch's last batches: the one synchronising check, where an epoch synchronises anyway
                blosum.check_fused(synchronize=True)

    def __getitem__(self, index):
        if isinstance(index, slice):
            s, e, st = index.indices(len(self))
            return self.__getitems__(list(range(s, e, st)))
        index = self._index(index)
        return self.get_batch(index, index + 1)[0]

    def fetch(self, index, return_items=False):
        """The reference's `fetch` (loaders.py:65-84) with its return shapes, on top of the batch encode:

        cnn=True   `index` a slice / index array / list -> (B, C, max_seq_len) float32 on the device ('length batch emb ->
                   batch emb length' + .float(), written directly); a single int -> the single-sequence one-hot
                   `tokenizer.onehot_encode(seq, padlen=max_seq_len)` as float32, (max(L, padlen) + bos + eos, C) -- no
                   rearrangement, as in the reference.  return_items=True -> `(tensor, items)` with the (augmented) sequences
                   as bytearrays.  (The reference returns None here unless return_items is set -- it only `return`s inside
                   `if return_items:`; this returns the tensor.)
        cnn=False  `index` an int: the token row (max_seq_len,) of that sequence (`return_items` is ignored, as in the reference)."""
        if not self.cnn:
            return self[index]
        single = not isinstance(index, (slice, list, tuple, np.ndarray, torch.Tensor))
        if single:
            i = self._index(index)
            chars, offs = self._packed_device(i, i + 1)
        elif isinstance(index, slice):
            s, e, st = index.indices(len(self))
            idx = list(range(s, e, st))
            contiguous = st == 1 and idx
            chars, offs = self._packed_device(s, e) if contiguous else self._packed_device(0, 0, idx)
        else:
            on_dev = isinstance(index, torch.Tensor) and index.is_cuda
            chars, offs = self._packed_device(0, 0, index if on_dev else [self._index(i) for i in np.asarray(index.cpu() if isinstance(index, torch.Tensor) else index).ravel()])
        if self.augment:
            self._calls += 1
            blosum.augment_packed(chars, offs, self.augment, self.augment_frac, self._seed + self._calls)
        items = None
        if single or return_items:
            ho = offs.cpu().numpy()
            hc = chars[:int(ho[-1])].cpu().numpy()
            items = [bytearray(hc[ho[k]:ho[k + 1]].tobytes()) for k in range(len(ho) - 1)]
        if single:
            ret = self.tokenizer.onehot_encode(items[0], padlen=self.max_seq_len, destchar="f", device=self.device)
            items = items[0]
        else:
            ret = self.tokenizer.onehot_packed(chars, offs, self.max_seq_len, "f", layout="bcl", validate=not self._trusted_lengths)
        return (ret, items) if return_items else ret

    def access(self, slc, stop=None, step=None):
        """Range accessor with the reference's convention (loaders.py:105-111): an int `slc` is the START of
        slice(slc, stop, step) -- so access(i) alone runs from i to the end --, a slice is used as it is."""
        rng_ = slc if isinstance(slc, slice) else slice(slc, stop, step)
        return self[rng_]

    def cleanup(self):
        """Nothing to release (the reference's loader closes its memmap here)."""


class AugmentedSeqDataset(FlatFileDataset):
    def __init__(self, ff, tokenizer, augment=1, augment_frac=.5, **kw):
        super().__init__(ff, tokenizer, augment=augment, augment_frac=augment_frac, **kw)
