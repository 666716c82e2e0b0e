"""ctypes binding of the C ABI (include/bsq.h) -- the same entry points the pybind11 layer calls.

Used by bench.py (kernel-only timing on raw device pointers), by the tests that exercise the ABI
directly, and as the worked example of INTEGRATION.md.  Raises if libbsq_hip.so is missing.
"""
from __future__ import annotations

import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbsq_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "bsq.h")
DIAG_HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "bsq_diag.h")

I8, I16, I32, U64, F32, F64 = range(6)
SPACE_HOST, SPACE_DEVICE = 0, 1
OK, ERR_INVALID_KEY, ERR_INVALID_ARG, ERR_DTYPE, ERR_SEQ_TOO_LONG, ERR_NO_DEVICE, ERR_HIP, ERR_ALLOC = range(8)


class Desc(ctypes.Structure):
    """struct bsq_desc"""
    _fields_ = [("lut", ctypes.c_int8 * 256), ("nchars", ctypes.c_int32), ("eos", ctypes.c_int32),
                ("bos", ctypes.c_int32), ("padchar", ctypes.c_int32)]


class Batch(ctypes.Structure):
    """struct bsq_batch: one packed batch of a multi-batch call (device pointers)"""
    _fields_ = [("chars", ctypes.c_void_p), ("offsets", ctypes.c_void_p), ("B", ctypes.c_int64), ("out", ctypes.c_void_p)]


_lib = None


def declared_symbols(header: str = None):
    """Every function name declared in include/bsq.h and include/bsq_diag.h (or in `header`)."""
    text = open(header).read() if header else open(HEADER_PATH).read() + open(DIAG_HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bsq_[a-z0-9_]+)\s*\(", text)))


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python bioseq_amd/build.py`")
    from . import _hipruntime
    _hipruntime.preload()
    L = ctypes.CDLL(LIB_PATH)
    c_int, i32, i64, vp, sz = ctypes.c_int, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_size_t
    dp = ctypes.POINTER(Desc)
    i64p = ctypes.POINTER(i64)
    sig = {
        "bsq_abi_version": (i32, []),
        "bsq_build_id": (ctypes.c_char_p, []),
        "bsq_strerror": (ctypes.c_char_p, [i32]),
        "bsq_last_error": (ctypes.c_char_p, []),
        "bsq_device_count": (i32, []),
        "bsq_tuning_set": (i32, [ctypes.c_char_p, i32]),
        "bsq_tuning_get": (i32, [ctypes.c_char_p]),
        "bsq_host_upload_bytes": (ctypes.c_uint64, []),
        "bsq_fused_status": (i32, [ctypes.POINTER(ctypes.c_uint32)]),
        "bsq_fused_status_clear": (None, []),
        "bsq_blosum62_accept_thresholds": (i32, [vp]),
        "bsq_num_keys": (i32, []),
        "bsq_key_name": (ctypes.c_char_p, [i32]),
        "bsq_lut_get": (i32, [ctypes.c_char_p, vp, ctypes.POINTER(i32)]),
        "bsq_desc_init": (i32, [dp, ctypes.c_char_p, i32, i32, i32]),
        "bsq_bos_id": (i32, [dp]),
        "bsq_eos_id": (i32, [dp]),
        "bsq_pad_id": (i32, [dp]),
        "bsq_alphabet_size": (i32, [dp]),
        "bsq_dtype_from_destchar": (i32, [ctypes.c_char, ctypes.POINTER(c_int)]),
        "bsq_dtype_size": (sz, [c_int]),
        "bsq_validate_lengths": (i32, [vp, i64, i64, i32, i32, i64p]),
        "bsq_validate_lengths_device": (i32, [vp, i64, i64, i32, i32, i64p, vp]),
        "bsq_validate_packed_device": (i32, [vp, i64, i64, i32, i32, i64, i64p, vp]),
        "bsq_xcd_round_robin": (i32, []),
        "bsq_tokenize_device": (i32, [dp, vp, vp, i64, i64, i32, c_int, vp, vp]),
        "bsq_tokenize_device_multi": (i32, [dp, i32, ctypes.POINTER(Batch), i64, i32, c_int, vp]),
        "bsq_onehot_device": (i32, [dp, vp, vp, vp, i64, i64, c_int, vp, vp]),
        "bsq_onehot_bcl_device": (i32, [dp, vp, vp, vp, i64, i64, c_int, vp, vp]),
        "bsq_onehot_block_device": (i32, [dp, vp, vp, vp, i64, i64, c_int, vp, i64, vp]),
        "bsq_tokenize_block_device": (i32, [dp, vp, vp, i64, i64, c_int, vp, i64, vp]),
        "bsq_onehot_kernel_name": (ctypes.c_char_p, [dp, i64, i64, c_int]),
        "bsq_tokenize_kernel_name": (ctypes.c_char_p, [dp, i64, i64, i32, c_int, i32]),
        "bsq_tokenize_device_generic": (i32, [dp, vp, vp, i64, i64, i32, c_int, vp, vp]),
        "bsq_onehot_device_generic": (i32, [dp, vp, vp, vp, i64, i64, c_int, vp, vp]),
        "bsq_fill_device": (i32, [vp, sz, ctypes.c_uint32, vp]),
        "bsq_fill_pattern_device": (i32, [vp, i64, i64, i32, i32, i32, i32, i32, vp]),
        "bsq_copy_mix_device": (i32, [vp, sz, vp, sz, i32, i32, vp]),
        "bsq_xcd_of_blocks_device": (i32, [vp, i32, vp]),
        "bsq_selftest_index_math": (i64, []),
        "bsq_raw_tokens_device": (i32, [vp, vp, vp, vp, i64, i64, vp, i64, vp]),
        "bsq_onehot_from_raw_tokens_device": (i32, [vp, i64, i64, i64, i32, i32, vp, vp]),
        "bsq_decode_sizes_device": (i32, [dp, vp, i32, i64, i64, i64, i64, vp, i64p, i64p, vp]),
        "bsq_decode_write_device": (i32, [dp, vp, i32, i64, i64, i64, i64, vp, vp, vp]),
        "bsq_argmax_tokens_device": (i32, [vp, i32, i64, i32, i64, vp, i32, vp]),
        "bsq_gather_packed_device": (i32, [vp, vp, i64, vp, i64, vp, i64, vp, vp, vp]),
        "bsq_blosum62_normrows": (i32, [vp]),
        "bsq_augment_device": (i32, [vp, vp, i64, i32, ctypes.c_double, ctypes.c_uint64, vp]),
        "bsq_augment_tokenize_device": (i32, [vp, vp, vp, i64, i64, i32, i32, vp, i32, ctypes.c_double, ctypes.c_uint64, vp]),
        "bsq_augment_device_multi": (i32, [i32, ctypes.POINTER(Batch), i32, ctypes.c_double, ctypes.POINTER(ctypes.c_uint64), vp]),
        "bsq_augment_tokenize_device_multi": (i32, [dp, i32, ctypes.POINTER(Batch), i64, i32, c_int, i32, ctypes.c_double,
                                                    ctypes.POINTER(ctypes.c_uint64), vp]),
        "bsq_tokenize_host": (i32, [dp, vp, vp, i64, i64, i32, c_int, vp, c_int, vp, i64p]),
        "bsq_onehot_host": (i32, [dp, vp, vp, vp, i64, i64, c_int, vp, c_int, vp, i64p]),
        "bsq_onehot_bcl_host": (i32, [dp, vp, vp, vp, i64, i64, c_int, vp, c_int, vp, i64p]),
        "bsq_stage_begin": (i32, [i64, sz, i32, vp, vp, vp, vp, vp]),
        "bsq_stage_upload": (i32, [vp, i64, i64, vp, vp, vp]),
        "bsq_stage_end": (i32, [vp]),
        "bsq_stage_result": (i32, [vp, sz, vp, vp]),
        "bsq_stage_fetch": (i32, [vp, sz, sz, vp]),
        "bsq_stage_wait": (i32, [vp, i32]),
        "bsq_stage_piece_hint": (i64, [i64, sz, sz, vp, vp, i64p]),
        "bsq_fastx_to_flatfile": (i32, [ctypes.c_char_p, ctypes.c_char_p, i64p, i64p]),
        "bsq_fastx_lengths": (i32, [ctypes.c_char_p, vp, i64, i64p]),
        "bsq_enable_peer_access": (i32, [i32, i32]),
        "bsq_pinned_scratch": (vp, [sz]),
        "bsq_release_staging": (None, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(status: int):
    if status != OK:
        L = load()
        raise RuntimeError(f"bsq status {status}: {L.bsq_strerror(status).decode()} -- {L.bsq_last_error().decode()}")


def make_desc(key: str, eos=False, bos=False, padchar=False) -> Desc:
    d = Desc()
    check(load().bsq_desc_init(ctypes.byref(d), key.encode(), int(bool(eos)), int(bool(bos)), int(bool(padchar))))
    return d


class _NoContext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NO_CONTEXT = _NoContext()


def on_device(device):
    """Context that makes `device` the current HIP device -- torch.cuda.device(device), but nothing at all when it is the current
    one already (the context manager and `torch.cuda.current_stream()` cost ~20 us of a loader batch's ~30: profiles/r04/loader_step_lab.txt)."""
    import torch
    idx = device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_CONTEXT
    return torch.cuda.device(device)


def raw_stream(device=None):
    """The current stream of `device` (default: the current device) as the integer the C ABI takes."""
    import torch
    idx = torch.cuda.current_device() if device is None or device.index is None else device.index
    fast = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if fast is not None:
        return fast(idx)
    return torch.cuda.current_stream(idx).cuda_stream
