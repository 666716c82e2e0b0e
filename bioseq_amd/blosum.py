"""BLOSUM62 substitution table and sequence augmentation -- the `bioseq.blosum` surface
(/root/reference/bioseq/blosum.py) with the batch path on the GPU.

`normrows` (21 x 20 float64, rows ARNDCQEGHILKMFPSTWYV + X, columns ARNDCQEGHILKMFPSTWYV) comes from
the native library and is bit-identical to the reference's numpy construction (blosum.py:36-45).

Batch augmentation -- what the reference's loaders do per sequence in Python right before tokenising
(bioseq/loaders.py:83,102; training/cnnpretrain.py:115-117) -- is `augment_packed`: in place on a
packed batch that already lives in HBM, one launch for the whole batch.  `substitute` / `augment_seq`
are kept as small host helpers with the reference's signatures.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import capi

_lib = capi.load()

# Residue order of the table: the 20 amino acids, then X (the row used for anything unknown).
true_aas = "ARNDCQEGHILKMFPSTWYV" + "X"
normrows = np.zeros((len(true_aas), len(true_aas) - 1), dtype=np.float64)
capi.check(_lib.bsq_blosum62_normrows(normrows.ctypes.data))
normrows.setflags(write=False)
aa_array = np.array([c for c in true_aas[:-1]])
ca = aa_array                                   # the reference exports both names
probdict = dict(zip(true_aas, (row.copy() for row in normrows)))
default_transitions = probdict["X"]
# Host helpers only.  Same seed as the reference (int(10000. / 137) == 72, blosum.py:6), but NOT the same stream: the
# reference draws 30 000 variates at import time (blosum.py:90-92) before any caller sees `rng`, and its accept loop
# consumes draws differently, so substitute() / augment_seq() agree with it in distribution, not draw for draw.
rng = np.random.default_rng(72)


def _row(residue):
    return probdict[residue] if residue in probdict else default_transitions


def substitute(inchar, size=1):
    """`size` replacement residues for `inchar`, drawn from its BLOSUM62 row (reference: blosum.py:51-60)."""
    return rng.choice(aa_array, size=size, replace=True, p=_row(inchar))


def augment_seq(inseq, chain_len=1):
    """`chain_len` point substitutions on one str, each at a uniformly drawn position and re-drawn (position included)
    until the residue really changes -- the host-side helper with the reference's semantics (blosum.py:63-87).
    Batches belong on the GPU: `augment_packed`."""
    residues = list(inseq)
    n = len(residues)
    for _step in range(chain_len):
        while True:
            where = int(rng.choice(n))
            new = str(substitute(residues[where])[0])
            if new != residues[where]:
                residues[where] = new
                break
    return "".join(residues)


def augment_packed(chars, offsets, chain_len=1, augment_frac=1.0, seed=0):
    """Mutate a packed batch IN PLACE on the GPU and return `chars`.

    chars / offsets: torch tensors on a HIP device (uint8[total], int64[B+1]).  Every sequence is
    augmented with probability `augment_frac` (>= 1: always -- the `augment_frac` of FlatFileDataset,
    loaders.py:35) by `chain_len` BLOSUM62-weighted point substitutions; unknown residues use the X row.
    Deterministic in (seed, sequence index).  Runs on torch's current stream.
    """
    import torch
    if not (chars.is_cuda and offsets.is_cuda):
        raise ValueError("augment_packed works on device tensors (use .to('cuda'))")
    if chars.dtype != torch.uint8 or offsets.dtype != torch.int64 or not chars.is_contiguous() or not offsets.is_contiguous():
        raise ValueError("chars must be contiguous uint8 and offsets contiguous int64")
    B = offsets.numel() - 1
    if chars.numel() == 0:
        return chars  # a batch of empty sequences: nothing to mutate
    with capi.on_device(chars.device):
        stream = capi.raw_stream(chars.device)
        capi.check(_lib.bsq_augment_device(chars.data_ptr(), offsets.data_ptr(), B, int(chain_len), float(augment_frac),
                                           ctypes.c_uint64(int(seed) & (2 ** 64 - 1)), stream))
    return chars


def check_fused(synchronize=False):
    """Raise RuntimeError if a token wave of a one-launch `augment_tokenize_packed` ever gave up waiting for its rows'
    augmentation (its part of the output is poisoned with 0xFF then; never observed -- see include/bsq.h).  Reads host memory
    only; covers the launches that have completed, so call it after a synchronisation, or pass `synchronize=True`.  The error
    is sticky: every later `augment_tokenize_packed` raises it too, until `clear_fused_error()`."""
    if synchronize:
        import torch
        torch.cuda.synchronize()
    capi.check(_lib.bsq_fused_status(None))


def clear_fused_error():
    _lib.bsq_fused_status_clear()


def augment_tokenize_packed(tokenizer, chars, offsets, padlen, destchar="b", batch_first=True, chain_len=1, augment_frac=1.0,
                            seed=0, out=None):
    """`augment_packed` followed by `tokenizer.tokenize_packed` -- what the reference's loaders do per item
    (bioseq/loaders.py:83-84: `augment_seq`, then `batch_tokenize`) -- with exactly their results: `chars` is mutated in
    place, the token matrix of the mutated batch is returned.  For `(B,P)` int8 matrices of the fast token kernel it is ONE
    launch (`bsq_augment_tokenize_device`); other shapes run the two launches.  Sequences longer than
    padlen - bos - eos must have been rejected by the caller (as `tokenize_packed(validate=True)` does).
    Raises RuntimeError (here, at the next call, or from `check_fused`) if an earlier one-launch call failed inside the kernel."""
    import torch
    if not (chars.is_cuda and offsets.is_cuda):
        raise ValueError("augment_tokenize_packed works on device tensors (use .to('cuda'))")
    if chars.dtype != torch.uint8 or offsets.dtype != torch.int64 or not chars.is_contiguous() or not offsets.is_contiguous():
        raise ValueError("chars must be contiguous uint8 and offsets contiguous int64")
    B = offsets.numel() - 1
    dt = ctypes.c_int(0)
    capi.check(_lib.bsq_dtype_from_destchar(destchar.encode(), ctypes.byref(dt)))
    tdt = {1: torch.int8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[_lib.bsq_dtype_size(dt)]
    if destchar[0].lower() == "f":
        tdt = torch.float32
    elif destchar[0].lower() == "d":
        tdt = torch.float64
    shape = (B, padlen) if batch_first else (padlen, B)
    if out is None:
        out = torch.empty(shape, dtype=tdt, device=chars.device)
    elif tuple(out.shape) != shape or out.dtype != tdt or not out.is_contiguous() or out.device != chars.device:
        raise ValueError("out must be a contiguous %s tensor of shape %r on the device of chars" % (tdt, shape))
    desc = capi.make_desc(tokenizer.key, tokenizer.includes_eos(), tokenizer.includes_bos(), tokenizer.is_padded())
    with capi.on_device(chars.device):
        stream = capi.raw_stream(chars.device)
        capi.check(_lib.bsq_augment_tokenize_device(ctypes.byref(desc), chars.data_ptr(), offsets.data_ptr(), B, int(padlen),
                                                    int(bool(batch_first)), dt, out.data_ptr(), int(chain_len) if chars.numel() else 0, float(augment_frac),
                                                    ctypes.c_uint64(int(seed) & (2 ** 64 - 1)), stream))
    return out


__all__ = ["aa_array", "substitute", "normrows", "probdict", "augment_seq", "augment_packed", "augment_tokenize_packed", "check_fused",
           "clear_fused_error"]
