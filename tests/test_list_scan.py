"""`cbioseq._ListScan` and `cbioseq._pack_list_into` -- the host half of `sharding.encode_on_devices` / `pack_once` ("host packs once; GPU g receives
its slice"): ONE scan of a Python list under the GIL (the item kinds of /root/reference/src/tokenize.h:292-322: str, bytes,
bytearray), offsets, the first over-long item, and ranges of it packed into memory the caller owns.  No device is needed for any of it."""
import numpy as np
import pytest


def _items(rng, n, hi):
    out = []
    for _ in range(n):
        raw = bytes(rng.integers(65, 91, size=int(rng.integers(0, hi + 1)), dtype=np.uint8))
        kind = int(rng.integers(0, 3))
        out.append(raw.decode() if kind == 0 else (raw if kind == 1 else bytearray(raw)))
    return out


def _as_bytes(x):
    return x.encode() if isinstance(x, str) else bytes(x)


@pytest.mark.parametrize("n,hi,nthreads", [(0, 10, 1), (1, 0, 1), (7, 30, 1), (1000, 200, 1), (20000, 40, 4), (20000, 40, 0), (3, 5000, 2)])
def test_scan_offsets_and_ranges(bsq, n, hi, nthreads):
    """offsets = the running sum of the items' byte lengths; pack(lo, hi, dst) writes exactly bytes [offsets[lo], offsets[hi]) of the
    concatenation and nothing else, for any cut of the list into ranges, in any order"""
    from bioseq_amd import cbioseq
    rng = np.random.default_rng(n * 31 + hi)
    items = _items(rng, n, hi)
    scan = cbioseq._ListScan(items, 1 << 40, nthreads)
    want = b"".join(_as_bytes(x) for x in items)
    lens = np.array([len(_as_bytes(x)) for x in items], dtype=np.int64)
    offs = np.asarray(scan.offsets)
    assert scan.n == n and scan.bad == -1 and offs.dtype == np.int64 and offs.shape == (n + 1,)
    assert offs[0] == 0 and (np.diff(offs) == lens).all() and int(offs[-1]) == len(want)
    cuts = sorted(set([0, n] + [int(c) for c in rng.integers(0, n + 1, size=5)]))
    dst = np.full(len(want) + 16, 0xEE, dtype=np.uint8)
    ranges = list(zip(cuts[:-1], cuts[1:]))
    for k in rng.permutation(len(ranges)):
        lo, hi_ = ranges[k]
        before = dst.copy()
        scan.pack(lo, hi_, dst)
        a, b = int(offs[lo]), int(offs[hi_])
        assert dst[a:b].tobytes() == want[a:b]
        assert (dst[:a] == before[:a]).all() and (dst[b:] == before[b:]).all()   # nothing outside the range is touched
    assert dst[:len(want)].tobytes() == want and (dst[len(want):] == 0xEE).all()
    scan.pack(0, 0, dst), scan.pack(n, n, dst)                                  # empty ranges are fine
    with pytest.raises(IndexError):
        scan.pack(0, n + 1, dst)
    with pytest.raises(IndexError):
        scan.pack(2, 1, dst)
    if len(want) > 0:
        with pytest.raises(ValueError):
            scan.pack(0, n, np.zeros(len(want) - 1, dtype=np.uint8))            # too small
        ro = np.zeros(len(want), dtype=np.uint8)
        ro.setflags(write=False)
        with pytest.raises((ValueError, BufferError)):
            scan.pack(0, n, ro)                                                  # not writable


def test_first_over_long_item_and_bad_items(bsq):
    """`bad` = the index of the FIRST item longer than maxlen (the caller raises the reference's error for it, tokenize.h:359-362 / :456-459);
    an item that is none of str / bytes / bytearray is the reference's ValueError (tokenize.h:316-321), raised by the scan itself"""
    from bioseq_amd import cbioseq
    items = ["ACGT", b"AC", bytearray(b"ACGTACGT"), "ACGTA", b"ACGTACGTACGT"]
    assert cbioseq._ListScan(items, 12).bad == -1
    assert cbioseq._ListScan(items, 7).bad == 2
    assert cbioseq._ListScan(items, 4).bad == 2
    assert cbioseq._ListScan(items, 1).bad == 0
    assert np.asarray(cbioseq._ListScan(items, 1).offsets).tolist() == [0, 4, 6, 14, 19, 31]   # the offsets are complete either way
    for wrong in ([b"AC", 5], [None], ["AC", 1.5], [["A"]]):
        with pytest.raises(ValueError, match="none of string, bytes"):
            cbioseq._ListScan(wrong, 100)
    with pytest.raises(TypeError):
        cbioseq._ListScan((x for x in items), 100)   # a generator is not a sequence (as in the reference)


@pytest.mark.parametrize("nthreads", [1, 3])
def test_pack_list_into_caller_memory(bsq, nthreads):
    """`_pack_list_into(batch, maxlen, nthreads, alloc)`: alloc(nbytes) is called ONCE, after the total is known, with total + 16 (spare bytes for
    the kernels' unaligned 16-byte loads); not at all when an item is too long"""
    from bioseq_amd import cbioseq
    rng = np.random.default_rng(5)
    items = _items(rng, 5000, 60)
    want = b"".join(_as_bytes(x) for x in items)
    asked = []

    def alloc(nbytes):
        asked.append(nbytes)
        return np.full(nbytes, 0xEE, dtype=np.uint8)

    offs, buf, bad = cbioseq._pack_list_into(items, 60, nthreads, alloc)
    assert bad == -1 and asked == [len(want) + 16] and buf[:len(want)].tobytes() == want and (buf[len(want):] == 0xEE).all()
    assert int(offs[-1]) == len(want) and offs.shape == (5001,)
    asked.clear()
    offs, buf, bad = cbioseq._pack_list_into(items + ["A" * 61] + items[:3], 60, nthreads, alloc)
    assert bad == 5000 and buf is None and asked == [] and offs.shape == (5005,)
    with pytest.raises(ValueError):
        cbioseq._pack_list_into(items, 60, nthreads, lambda nbytes: np.zeros(nbytes - 17, dtype=np.uint8))   # alloc() returned too little
