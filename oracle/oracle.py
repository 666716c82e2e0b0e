"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of the CPU oracle (oracle/bsq_oracle.c).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
bioseq_amd/ never imports this module (tests/test_abi_and_layout.py greps for that).

``OracleTokenizer`` mirrors the Python surface of the reference's ``cbioseq.Tokenizer``
(/root/reference/src/tokenize.cpp:22-112) for the batch entry points so parity tests can call
the oracle, the compiled reference (oracle/_ref) and the HIP product with the same arguments.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libbsq_oracle.so")
_lib = None

I8, I16, I32, U64, F32, F64 = range(6)
NP_DTYPES = {I8: np.int8, I16: np.int16, I32: np.int32, U64: np.uint64, F32: np.float32, F64: np.float64}


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "bsq_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "port"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        i64, ci, vp = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p
        L.bsqo_alphabet.argtypes = [ctypes.c_char_p, vp, ctypes.POINTER(ci)]
        L.bsqo_alphabet.restype = ci
        L.bsqo_dtype_from_destchar.argtypes = [ctypes.c_char]
        L.bsqo_dtype_from_destchar.restype = ci
        L.bsqo_key.restype = ctypes.c_char_p
        L.bsqo_key.argtypes = [ci]
        L.bsqo_tokenize.argtypes = [vp, ci, ci, ci, ci, vp, vp, i64, i64, ci, ci, vp, ci]
        L.bsqo_tokenize.restype = i64
        L.bsqo_onehot.argtypes = [vp, ci, ci, ci, ci, vp, vp, vp, i64, i64, ci, vp, ci]
        L.bsqo_onehot.restype = i64
        _lib = L
    return _lib


def keys():
    L = lib()
    return [L.bsqo_key(i).decode() for i in range(L.bsqo_num_keys())]


def pack(batch):
    """list of str/bytes/bytearray -> (chars uint8[total], offsets int64[B+1])."""
    items = [s.encode("utf-8") if isinstance(s, str) else bytes(s) for s in batch]
    offsets = np.zeros(len(items) + 1, dtype=np.int64)
    if items:
        np.cumsum([len(x) for x in items], out=offsets[1:])
    chars = np.frombuffer(b"".join(items), dtype=np.uint8).copy()
    return chars, offsets


def pack_mask(mask, offsets):
    """Reference mask form (list with one uint8 array or None per sequence,
    /root/reference/src/tokenize.h:294-297) -> one byte per input character, or None."""
    if not isinstance(mask, list):
        return None  # anything that is not a list is silently ignored by the reference
    out = np.ones(int(offsets[-1]), dtype=np.uint8)
    for i, m in enumerate(mask):
        if isinstance(m, np.ndarray):
            n = int(offsets[i + 1] - offsets[i])
            out[offsets[i]:offsets[i + 1]] = np.asarray(m, dtype=np.uint8).ravel()[:n]
    return out


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class OracleTokenizer:
    def __init__(self, key, eos=False, bos=False, padchar=False):
        L = lib()
        self.lut_arr = np.empty(256, dtype=np.int8)
        n = ctypes.c_int(0)
        if L.bsqo_alphabet(key.encode(), _ptr(self.lut_arr), ctypes.byref(n)) != 0:
            raise RuntimeError("Invalid tokenizer type; select one from" + "".join(k + ";" for k in keys()))
        self.key = key.upper()
        self._nchars = n.value
        self._eos, self._bos, self._pad = int(bool(eos)), int(bool(bos)), int(bool(padchar))

    # ids (reference src/tokenize.h:22-38)
    def nchars(self): return self._nchars
    def bos(self): return lib().bsqo_bos_id(self._nchars, self._eos, self._bos, self._pad)
    def eos(self): return lib().bsqo_eos_id(self._nchars, self._eos, self._bos, self._pad)
    def pad(self): return lib().bsqo_pad_id(self._nchars, self._eos, self._bos, self._pad)
    def alphabet_size(self): return lib().bsqo_alphabet_size(self._nchars, self._eos, self._bos, self._pad)
    def is_padded(self): return bool(self._pad)
    def includes_bos(self): return bool(self._bos)
    def includes_eos(self): return bool(self._eos)

    @staticmethod
    def _dtype(destchar):
        d = lib().bsqo_dtype_from_destchar(destchar[0].encode("latin-1"))
        if d < 0:
            raise ValueError("Unsupported dtype: " + destchar)
        return d

    def tokenize_packed(self, chars, offsets, padlen, destchar="B", batch_first=False, nthreads=1):
        if padlen <= 0:
            raise ValueError("batch tokenize requires padlen is provded.")
        d = self._dtype(destchar)
        B = len(offsets) - 1
        out = np.empty((B, padlen) if batch_first else (padlen, B), dtype=NP_DTYPES[d])
        chars = np.ascontiguousarray(chars, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        rc = lib().bsqo_tokenize(_ptr(self.lut_arr), self._nchars, self._eos, self._bos, self._pad,
                                 _ptr(chars), _ptr(offsets), B, padlen, int(batch_first), d, _ptr(out), nthreads)
        if rc:
            tl = int(offsets[rc] - offsets[rc - 1]) + self._bos + self._eos
            raise ValueError(f"seq len + bos + eos > padlen: {tl}, vs padlen {padlen}")
        return out

    def onehot_packed(self, chars, offsets, padlen, destchar="B", nthreads=1, mask=None):
        if padlen <= 0:
            raise ValueError("batch tokenize requires padlen is provded.")
        d = self._dtype(destchar)
        B = len(offsets) - 1
        out = np.empty((padlen, B, self.alphabet_size()), dtype=NP_DTYPES[d])
        chars = np.ascontiguousarray(chars, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
        rc = lib().bsqo_onehot(_ptr(self.lut_arr), self._nchars, self._eos, self._bos, self._pad,
                               _ptr(chars), _ptr(offsets), _ptr(mask), B, padlen, d, _ptr(out), nthreads)
        if rc:
            tl = int(offsets[rc] - offsets[rc - 1]) + self._bos + self._eos
            raise ValueError(f"seq len + bos + eos > padlen: {tl}, vs padlen {padlen}")
        return out

    # reference-shaped entry points
    def batch_tokenize(self, batch, padlen=-1, destchar="B", batch_first=False, nthreads=1):
        chars, offsets = pack(batch)
        return self.tokenize_packed(chars, offsets, padlen, destchar, batch_first, nthreads)

    def batch_onehot_encode(self, batch, padlen=-1, destchar="B", nthreads=1, mask=None):
        chars, offsets = pack(batch)
        return self.onehot_packed(chars, offsets, padlen, destchar, nthreads, pack_mask(mask, offsets))


def load_reference():
    """Return the compiled reference module (oracle/_ref/cbioseq*.so) or None when absent."""
    import glob
    import importlib.util
    hits = glob.glob(os.path.join(_HERE, "_ref", "cbioseq*.so"))
    if not hits:
        return None
    spec = importlib.util.spec_from_file_location("cbioseq", hits[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
