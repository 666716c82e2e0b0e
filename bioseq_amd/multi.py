"""Several independent packed batches per call (round 6; `bsq_tokenize_device_multi`, `bsq_augment_tokenize_device_multi`).

The reference encodes one batch per call and its training loop issues the calls back to back (/root/reference/bioseq/loaders.py:76-104;
`Tokenizer::transencode`, src/tokenize.h:451-479, is one OpenMP region per batch).  On the GPU a 16-40-us token launch pays its own ramp-up
and drain, and on one in-order stream the next batch cannot start under the tail of this one; a caller that has its next batches at hand
passes them together and gets ONE launch (two with augmentation) for up to eight of them.  Results are bit for bit those of the per-batch
calls `tok.tokenize_packed(...)` / `blosum.augment_tokenize_packed(...)` with the same seeds."""
from __future__ import annotations

import ctypes

from . import capi

_TORCH_DTYPES = None


def _dtype(code):
    import torch
    global _TORCH_DTYPES
    if _TORCH_DTYPES is None:
        _TORCH_DTYPES = {0: torch.int8, 1: torch.int16, 2: torch.int32, 3: torch.int64, 4: torch.float32, 5: torch.float64}
    return _TORCH_DTYPES[code]


def _prepare(tokenizer, batches, padlen, destchar, batch_first, outs):
    import torch
    lib = capi.load()
    desc = capi.make_desc(tokenizer.key, tokenizer.includes_eos(), tokenizer.includes_bos(), tokenizer.is_padded())
    dt = ctypes.c_int(0)
    capi.check(lib.bsq_dtype_from_destchar(destchar.encode(), ctypes.byref(dt)))
    tdt = _dtype(dt.value)
    n = len(batches)
    arr = (capi.Batch * max(n, 1))()
    results, keep = [], []
    dev = None
    room = int(padlen) - int(tokenizer.includes_bos()) - int(tokenizer.includes_eos())
    for i, (chars, offsets) in enumerate(batches):
        if not (isinstance(chars, torch.Tensor) and chars.is_cuda and isinstance(offsets, torch.Tensor) and offsets.is_cuda):
            raise ValueError("the multi-batch calls work on packed batches resident on the device (chars, offsets tensors)")
        if dev is None:
            dev = chars.device
        if chars.device != dev or offsets.device != dev:
            raise ValueError("every batch of a multi-batch call lives on one device")
        offsets = offsets.to(torch.int64).contiguous()
        chars = chars.contiguous()
        B = int(offsets.shape[0]) - 1
        shape = (B, padlen) if batch_first else (padlen, B)
        if outs is not None:
            out = outs[i]
            if tuple(out.shape) != shape or out.dtype != tdt or not out.is_contiguous() or out.device != dev:
                raise ValueError("outs[%d] must be a contiguous %s tensor of shape %r on %s" % (i, tdt, shape, dev))
        else:
            out = torch.empty(shape, dtype=tdt, device=dev)
        if chars.numel() == 0 and B > 0:
            # every sequence of this batch is empty: torch hands out a null data_ptr for a tensor without elements, which the augmentation entry
            # points refuse for B > 0 (include/bsq.h) -- no kernel reads a character of an empty sequence, so any valid address will do
            chars = torch.zeros(16, dtype=torch.uint8, device=dev)
        keep.append((chars, offsets))
        results.append(out)
        arr[i].chars, arr[i].offsets, arr[i].B, arr[i].out = chars.data_ptr(), offsets.data_ptr(), B, out.data_ptr()
    return lib, desc, dt, arr, results, keep, dev, room


def validate_packed_multi(tokenizer, batches, padlen):
    """The reference's over-long-sequence error for every batch (one synchronising check per batch: `tokenize_packed(validate=True)`'s)."""
    lib = capi.load()
    desc = capi.make_desc(tokenizer.key, tokenizer.includes_eos(), tokenizer.includes_bos(), tokenizer.is_padded())
    for chars, offsets in batches:
        B = int(offsets.shape[0]) - 1
        if B <= 0:
            continue
        bad = ctypes.c_int64(-1)
        with capi.on_device(chars.device):
            st = lib.bsq_validate_packed_device(offsets.data_ptr(), B, padlen, desc.bos, desc.eos, chars.numel(), ctypes.byref(bad),
                                                ctypes.c_void_p(capi.raw_stream(chars.device)))
        if st == capi.ERR_SEQ_TOO_LONG:  # the reference's error of batch_tokenize (tokenize.h:456-459), as tokenize_packed raises it
            i = int(bad.value)
            length = int(offsets[i + 1] - offsets[i]) + int(desc.bos) + int(desc.eos)
            raise RuntimeError("seq len + bos + eos > padlen: %d, vs padlen %d" % (length, int(padlen)))
        capi.check(st)


def tokenize_packed_multi(tokenizer, batches, padlen, destchar="B", batch_first=False, outs=None, validate=True):
    """`[tokenizer.tokenize_packed(c, o, padlen, destchar, batch_first) for c, o in batches]` in ceil(n / 8) launches.
    batches: list of (chars uint8, offsets int64) device tensors; returns the list of token matrices ((B_i, padlen) or (padlen, B_i))."""
    if validate:
        validate_packed_multi(tokenizer, batches, padlen)
    lib, desc, dt, arr, results, keep, dev, _ = _prepare(tokenizer, batches, padlen, destchar, batch_first, outs)
    if batches:
        with capi.on_device(dev):
            capi.check(lib.bsq_tokenize_device_multi(ctypes.byref(desc), len(batches), arr, padlen, int(batch_first), dt, ctypes.c_void_p(capi.raw_stream(dev))))
    return results


def augment_tokenize_packed_multi(tokenizer, batches, padlen, destchar="b", batch_first=True, chain_len=1, augment_frac=1.0, seeds=None, outs=None,
                                  validate=True):
    """`[blosum.augment_tokenize_packed(tokenizer, c, o, padlen, destchar, batch_first, chain_len, augment_frac, seed) for ...]`: every batch's
    characters are mutated in place (BLOSUM62 point substitutions, `seeds[i]` is batch i's seed), the token matrices of the mutated batches
    come back -- two launches for up to eight batches instead of one or two per batch."""
    n = len(batches)
    seeds = list(range(n)) if seeds is None else [int(s) for s in seeds]
    if len(seeds) != n:
        raise ValueError("one seed per batch")
    if validate:
        validate_packed_multi(tokenizer, batches, padlen)
    lib, desc, dt, arr, results, keep, dev, _ = _prepare(tokenizer, batches, padlen, destchar, batch_first, outs)
    if batches:
        sd = (ctypes.c_uint64 * n)(*[s & ((1 << 64) - 1) for s in seeds])
        with capi.on_device(dev):
            capi.check(lib.bsq_augment_tokenize_device_multi(ctypes.byref(desc), n, arr, padlen, int(batch_first), dt, int(chain_len), float(augment_frac),
                                                             sd, ctypes.c_void_p(capi.raw_stream(dev))))
    return results
