"""FlatFile: the reference's random-access sequence store as a ZERO-COPY batch source.

On-disk format (/root/reference/src/fxstats.cpp:33-64, reader :66-75):

    uint64 nseqs | uint64 offsets[nseqs + 1] | sequence bytes, concatenated

which is exactly the packed batch (`chars`, `offsets`) the device kernels consume.  The reference
materialises a `bytearray` per sequence (`access`, fxstats.cpp:128-133) and the tokenizer then
re-extracts the pointers; here `packed(start, stop)` hands a slice of the memory-mapped file straight
to `Tokenizer.tokenize_packed` / `onehot_packed`, and `to_device()` uploads the whole store once so
that batches are encoded from HBM with no per-batch host work at all.

Same Python surface as `cbioseq.FlatFile` (fxstats.cpp:166-200): `FlatFile(path, maxseqlen=-1)` opens
a `.ff`, `FlatFile(fastx, out)` builds it from FASTA/FASTQ (optionally gzipped) first; `access`,
`__getitem__` (int / slice / index array), `nseqs()`, `size()`, `len()`, `seq_offset()`, `indptr()`,
`maxseqlen` / `max_seq_len`, `path`.
"""
from __future__ import annotations

import os

import numpy as np


def fastx_to_flatfile(inpath, outpath=""):
    """FASTA / FASTQ (plain or gzip) -> FlatFile at `outpath` (default inpath + '.ff'), by the native streaming reader
    (`bsq_fastx_to_flatfile`, csrc/bsq_fastx.cpp: the reference's kseq record grammar, only the offsets held in memory).
    Returns (outpath, nseqs, max_seq_len) -- the reference's FlatFile::make (fxstats.cpp:33-64)."""
    import ctypes
    from . import capi
    lib = capi.load()
    outpath = outpath or inpath + ".ff"
    n, longest = ctypes.c_int64(0), ctypes.c_int64(0)
    st = lib.bsq_fastx_to_flatfile(os.fsencode(inpath), os.fsencode(outpath), ctypes.byref(n), ctypes.byref(longest))
    if st:
        raise RuntimeError(lib.bsq_last_error().decode() or "bsq_fastx_to_flatfile failed")   # std::runtime_error in the reference
    return outpath, n.value, longest.value


def getstats(paths):
    """Sequence lengths of every record of every FASTA / FASTQ file in `paths`: a list of uint64 arrays -- the reference's
    module-level `getstats` (fxstats.cpp:12-23, 202-219)."""
    import ctypes
    from . import capi
    lib = capi.load()
    out = []
    for path in paths:
        n = ctypes.c_int64(0)
        cap = 1 << 16
        while True:
            lens = np.empty(cap, dtype=np.uint64)
            st = lib.bsq_fastx_lengths(os.fsencode(path), lens.ctypes.data, cap, ctypes.byref(n))
            if st:
                raise RuntimeError(lib.bsq_last_error().decode() or "bsq_fastx_lengths failed")
            if n.value <= cap:
                break
            cap = n.value
        out.append(lens[:n.value].copy())
    return out


def write_flatfile(seqs, path):
    """Write sequences (bytes / str) in the FlatFile format; returns `path`."""
    items = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
    offsets = np.zeros(len(items) + 1, dtype="<u8")
    if items:
        np.cumsum([len(x) for x in items], out=offsets[1:])
    with open(path, "wb") as f:
        f.write(np.array([len(items)], dtype="<u8").tobytes())
        f.write(offsets.tobytes())
        for x in items:
            f.write(x)
    return path


class FlatFileIterator:
    """What iterating a FlatFile yields -- the reference's `cbioseq.FlatFileIterator` (fxstats.cpp:136-160, bound at :163-168):
    an object with the read-only properties `.seq` / `.sequence` (a fresh bytearray of the sequence it stands on), itself
    iterable; `next()` advances and hands back a snapshot of the new position."""
    __slots__ = ("_ff", "_pos", "_stop")

    def __init__(self, ff, pos=-1, stop=None):
        if isinstance(ff, FlatFileIterator):  # the reference's copy constructor (py::init<FlatFileIterator>)
            ff, pos, stop = ff._ff, ff._pos, ff._stop
        self._ff, self._pos = ff, int(pos)
        self._stop = ff.nseqs() if stop is None else int(stop)

    def __iter__(self):
        return FlatFileIterator(self)

    def __next__(self):
        self._pos += 1
        if self._pos >= self._stop:
            raise StopIteration("End of iterator")
        return FlatFileIterator(self)

    @property
    def sequence(self):
        return self._ff.access(self._pos)

    seq = sequence


class FlatFile:
    # The reference's iterator starts ON sequence 0 and pre-increments in __next__ (fxstats.cpp:143-146): `for x in ff` there
    # never yields sequence 0 (probed on the compiled reference: 5 sequences -> 4 items; one sequence -> none).  Here iteration
    # yields every sequence; set this to True to reproduce the reference's off-by-one bit for bit (difference 10).
    ITER_SKIPS_FIRST = False

    def __init__(self, inputfile, maxseqlen=-1):
        """FlatFile(path_to_ff, maxseqlen=-1) opens a store; FlatFile(fastx_path, out_path) builds one first
        (`out_path == ''` -> fastx_path + '.ff'), as the reference's two constructors do."""
        if isinstance(maxseqlen, str):
            inputfile, _, _ = fastx_to_flatfile(inputfile, maxseqlen)
            maxseqlen = -1
        self.path = inputfile
        self._mm = np.memmap(inputfile, mode="r", dtype=np.uint8) if os.path.getsize(inputfile) else np.zeros(8, np.uint8)
        self._n = int(self._mm[:8].view("<u8")[0])
        self._seq_offset = (self._n + 2) * 8
        self._offsets = self._mm[8:self._seq_offset].view("<u8").astype(np.int64)  # small copy: B+1 entries
        self._chars = self._mm[self._seq_offset:self._seq_offset + int(self._offsets[-1])]
        self._longest = int(np.diff(self._offsets).max()) if self._n else 0   # the true maximum, whatever maxseqlen says
        if maxseqlen is None or maxseqlen < 0:
            maxseqlen = self._longest
        self._maxseqlen = int(maxseqlen)
        self._dev = {}
        self._top_cum = None

    # ---- reference surface ------------------------------------------------------------------
    def nseqs(self):
        return self._n

    size = nseqs

    def __len__(self):
        return self._n

    def seq_offset(self):
        return self._seq_offset

    def indptr(self):
        return self._offsets.astype(np.uint64)

    @property
    def maxseqlen(self):
        return self._maxseqlen

    max_seq_len = maxseqlen

    def _one(self, i):
        if i < 0 or i >= self._n:
            raise IndexError("Accessing sequence out of range")
        return bytearray(self._chars[self._offsets[i]:self._offsets[i + 1]].tobytes())

    def access(self, start, stop=None, step=1):
        if isinstance(start, slice):
            return [self._one(i) for i in range(*start.indices(self._n))]
        if stop is None:
            return self._one(int(start))
        if step == 0:
            raise ValueError("step must be nonzero")
        return [self._one(i) for i in range(int(start), int(stop), int(step))]

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            return self.access(idx)
        if isinstance(idx, (np.ndarray, list, tuple)):
            return [self[int(i)] for i in np.asarray(idx).ravel()]
        idx = int(idx)
        if idx < 0:
            if idx < -self._n:
                raise IndexError("For a negative index, idx must be >= -len(x)")
            idx += self._n
        return self._one(idx)

    def __iter__(self):
        """FlatFileIterator objects (`.seq` / `.sequence`), as the reference yields them (fxstats.cpp:177) -- not bare bytearrays."""
        return FlatFileIterator(self, 0 if (self.ITER_SKIPS_FIRST and self._n) else -1)

    # ---- zero-copy batch source -------------------------------------------------------------
    def packed(self, start=0, stop=None):
        """(chars uint8 view into the mapped file, offsets int64 rebased to 0) of sequences [start, stop)."""
        stop = self._n if stop is None else min(int(stop), self._n)
        start = max(0, int(start))
        offs = self._offsets[start:stop + 1]
        return self._chars[offs[0]:offs[-1]], offs - offs[0]

    def to_device(self, device="cuda"):
        """Upload the whole store once; returns (chars, offsets) device tensors (cached per device)."""
        import torch
        dev = torch.device(device)
        if dev not in self._dev:
            self._dev[dev] = (torch.from_numpy(np.array(self._chars, copy=True)).to(dev),
                              torch.from_numpy(self._offsets).to(dev))
        return self._dev[dev]

    def packed_device(self, start=0, stop=None, device="cuda"):
        """Device-resident packed batch of sequences [start, stop): views of the uploaded store + a
        rebased offsets tensor -- no host work, no copy of the characters."""
        chars, offs = self.to_device(device)
        stop = self._n if stop is None else min(int(stop), self._n)
        o = offs[start:stop + 1]
        c0, c1 = int(self._offsets[start]), int(self._offsets[stop])
        return chars[c0:c1], o - o[0]

    def gather_device(self, indices, device="cuda", validate=True, distinct=False):
        """Packed batch (chars, offsets) of the sequences `indices` -- any order, repeats allowed -- rebuilt ON THE
        DEVICE from the uploaded store (`bsq_gather_packed_device`): what a shuffling sampler needs, with no host gather
        and no upload of characters.  `indices`: an int64 tensor already on `device` (nothing crosses PCIe at all) or a
        host list / array (range-checked here, 8 bytes per index uploaded).  The returned `chars` tensor may be longer
        than the batch (device indices: sized by a bound); only its first offsets[-1] bytes belong to it.
        validate=False: a device index tensor the caller vouches for (in range) -- nothing is read back, the call never
        synchronises; with distinct=True (no index twice: a sampler's permutation) the buffer is sized by the n longest
        sequences of the store instead of n times the longest."""
        import ctypes
        import torch
        from . import capi
        lib = capi.load()
        dev = torch.device(device)
        chars, offs = self.to_device(dev)
        on_device = isinstance(indices, torch.Tensor) and indices.is_cuda
        if on_device:
            idx = indices.to(torch.int64).contiguous()
            if idx.device != chars.device:
                raise ValueError("indices live on %s, the store on %s" % (idx.device, chars.device))
        else:
            host = np.ascontiguousarray(np.asarray(indices, dtype=np.int64).ravel())
            if host.size and (host.min() < 0 or host.max() >= self._n):
                raise IndexError("Accessing sequence out of range")
            idx = torch.from_numpy(host).to(dev)
        n = idx.numel()
        # Size of the batch's characters.  Host indices: exact (the lengths are known here).  Device indices: the n longest
        # sequences of the store bound every list WITHOUT repeats (a sampler's permutation) -- one 35 000-residue outlier no longer
        # makes every 4096-sequence batch 143 MB --; a list with repeats can exceed that, the kernel reports the overflow, and the
        # call is repeated with the unconditional bound n * longest (unchecked lists, validate=False, start there).
        if not on_device:
            capacity = int((self._offsets[host + 1] - self._offsets[host]).sum()) if n else 0
        elif (validate or distinct) and n <= self._n:
            capacity = int(self._top_lengths_cumsum()[n - 1]) if n else 0
        else:
            capacity = n * self._longest
        out_offs = torch.empty(n + 1, dtype=torch.int64, device=dev)
        status = torch.empty(1, dtype=torch.int64, device=dev) if (on_device and validate) else None
        while True:
            out_chars = torch.empty(max(capacity, 1), dtype=torch.uint8, device=dev)
            with capi.on_device(dev):
                stream = ctypes.c_void_p(capi.raw_stream(dev))
                capi.check(lib.bsq_gather_packed_device(chars.data_ptr(), offs.data_ptr(), self._n, idx.data_ptr(), n,
                                                        out_chars.data_ptr(), capacity, out_offs.data_ptr(),
                                                        status.data_ptr() if status is not None else None, stream))
            if not (on_device and validate):  # (host lists were range-checked above and sized exactly)
                break
            bad = int(status.item())  # the only synchronising step, and only for index tensors nobody has checked
            if bad < 0:
                break
            if bad >= n and capacity < n * self._longest:  # overflow of the no-repeats bound: the list repeats long sequences
                capacity = n * self._longest
                continue
            raise IndexError("Accessing sequence out of range (position %d of the index list)" % (bad % max(n, 1)))
        return (out_chars[:capacity] if not on_device else out_chars), out_offs

    def _top_lengths_cumsum(self):
        """cumsum of the store's sequence lengths in descending order: entry n - 1 bounds the characters of any n distinct sequences."""
        if self._top_cum is None:
            self._top_cum = np.cumsum(np.sort(np.diff(self._offsets))[::-1], dtype=np.int64)
        return self._top_cum

    def batch_tokenize(self, tokenizer, start=0, stop=None, padlen=None, destchar="B", batch_first=True, device=None):
        """`tokenizer.batch_tokenize(ff.access(start, stop), ...)` without materialising the sequences
        (cf. FF2NP, bioseq/loaders.py:11-26).  padlen defaults to maxseqlen + bos + eos."""
        if padlen is None:
            padlen = self._maxseqlen + tokenizer.includes_bos() + tokenizer.includes_eos()
        if device is not None:
            c, o = self.packed_device(start, stop, device)
        else:
            c, o = self.packed(start, stop)
        return tokenizer.tokenize_packed(c, o, padlen, destchar, batch_first)

    def batch_onehot_encode(self, tokenizer, start=0, stop=None, padlen=None, destchar="B", device=None):
        if padlen is None:
            padlen = self._maxseqlen + tokenizer.includes_bos() + tokenizer.includes_eos()
        if device is not None:
            c, o = self.packed_device(start, stop, device)
        else:
            c, o = self.packed(start, stop)
        return tokenizer.onehot_packed(c, o, padlen, destchar)
