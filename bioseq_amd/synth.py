"""Deterministic synthetic sequence batches (SURVEY.md Appendix A generator).

Counter-based splitmix64: every (sequence, position) is an independent function of
``(seed, i, j)``, so a batch can be produced in any order, in chunks, or per rank.
The output is the *packed* form the device kernels consume:

    chars   : uint8[total]   all residues, concatenated
    offsets : int64[B + 1]   sequence i is chars[offsets[i]:offsets[i+1]]

This is the same CSR layout as the reference's on-disk FlatFile
(/root/reference/src/fxstats.cpp:33-64), which is why it is the kernels' native input.
"""
from __future__ import annotations

import numpy as np

GAMMA = np.uint64(0x9E3779B97F4A7C15)
C1 = np.uint64(0xD1342543DE82EF95)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

AA = "ACDEFGHIKLMNPQRSTVWY"
DIRTY = AA + AA.lower() + "XBZUO*-nN"


def _mix(z: np.ndarray) -> np.ndarray:
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def synth_lengths(seed: int, n: int, lo: int, hi: int, first: int = 0) -> np.ndarray:
    """Lengths of sequences ``first .. first+n-1`` of the stream ``seed``."""
    with np.errstate(over="ignore"):
        i = np.arange(first, first + n, dtype=np.uint64)
        base = np.uint64(seed) + i * C1
        return (np.uint64(lo) + _mix(base + GAMMA) % np.uint64(hi - lo + 1)).astype(np.int64)


def synth_packed(seed: int, n: int, lo: int, hi: int, letters: str, first: int = 0,
                 chunk: int = 1 << 18):
    """Return ``(chars uint8[total], offsets int64[n+1])`` for sequences first..first+n-1."""
    lens = synth_lengths(seed, n, lo, hi, first)
    offsets = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    total = int(offsets[-1])
    chars = np.empty(total, dtype=np.uint8)
    table = np.frombuffer(letters.encode("ascii"), dtype=np.uint8)
    nl = np.uint64(len(letters))
    with np.errstate(over="ignore"):
        base = np.uint64(seed) + np.arange(first, first + n, dtype=np.uint64) * C1
        # walk the batch in blocks of whole sequences holding <= chunk characters
        b0 = 0
        while b0 < n:
            b1 = int(np.searchsorted(offsets, offsets[b0] + chunk, side="right")) - 1
            b1 = min(n, max(b1, b0 + 1))
            s, e = int(offsets[b0]), int(offsets[b1])
            if e > s:
                ln = lens[b0:b1]
                j = np.arange(e - s, dtype=np.uint64) - np.repeat((offsets[b0:b1] - s).astype(np.uint64), ln)
                z = _mix(np.repeat(base[b0:b1], ln) + (j + np.uint64(2)) * GAMMA)
                chars[s:e] = table[(z % nl).astype(np.int64)]
            b0 = b1
    return chars, offsets


def unpack(chars: np.ndarray, offsets: np.ndarray, as_str: bool = False):
    """Packed batch -> list of ``bytes`` (or ``str``): the reference API's input form."""
    buf = chars.tobytes()
    off = offsets.tolist()
    if as_str:
        s = buf.decode("latin-1")
        return [s[off[i]:off[i + 1]] for i in range(len(off) - 1)]
    return [buf[off[i]:off[i + 1]] for i in range(len(off) - 1)]


def synth(seed: int, n: int, lo: int, hi: int, letters: str):
    """List-of-str form (small batches / doc examples)."""
    c, o = synth_packed(seed, n, lo, hi, letters)
    return unpack(c, o, as_str=True)


# BASELINE.json configs.  key, (eos,bos,padchar) in ctor order.
CONFIGS = {
    "cfg1": dict(seed=101, n=1000, lo=1, hi=254, letters="ACGT", key="DNA", eos=1, bos=1, padchar=1,
                 padlen=256),
    "cfg2": dict(seed=202, n=65536, lo=50, hi=1024, letters=AA, key="AMINO20", eos=0, bos=0, padchar=0,
                 padlen=1024),
    "cfg4": dict(seed=404, n=1000000, lo=150, hi=150, letters="ACGT", key="DNA4", eos=1, bos=1,
                 padchar=1, padlen=160),
    "cfg5": dict(seed=505, n=262144, lo=30, hi=512, letters=AA, key="SEB8", eos=0, bos=0, padchar=0,
                 padlen=512),
    "dirty": dict(seed=606, n=2048, lo=0, hi=200, letters=DIRTY, padlen=202),
}
CONFIGS["cfg3"] = CONFIGS["cfg2"]
