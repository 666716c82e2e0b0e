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

true_aas = 'ARNDCQEGHILKMFPSTWYVX'
_lib = capi.load()
normrows = np.zeros((21, 20), dtype=np.float64)
capi.check(_lib.bsq_blosum62_normrows(normrows.ctypes.data))
normrows.setflags(write=False)
ca = np.array(list(true_aas))[:-1]
aa_array = ca
probdict = {k: normrows[idx].copy() for idx, k in enumerate(true_aas)}
default_transitions = probdict['X']
rng = np.random.default_rng(int(10000. / 137))


def substitute(inchar, size=1):
    """Sample `size` replacement residues for `inchar` from the BLOSUM62 row (blosum.py:51-60)."""
    return rng.choice(ca, p=probdict.get(inchar, default_transitions), size=size, replace=True)


def augment_seq(inseq, chain_len=1):
    """Mutate one sequence `chain_len` times on the host (blosum.py:63-87)."""
    ls = len(inseq)
    for _ in range(chain_len):
        outchar, inchar = (0, 0)
        while inchar == outchar:
            idx = rng.choice(ls)
            outchar = inseq[idx]
            inchar = substitute(outchar)[0]
        ba = bytearray(inseq, 'utf-8')
        ba[idx] = ord(inchar)
        inseq = ba.decode()
    return inseq


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
    with torch.cuda.device(chars.device):
        stream = torch.cuda.current_stream().cuda_stream
        capi.check(_lib.bsq_augment_device(chars.data_ptr(), offsets.data_ptr(), B, int(chain_len), float(augment_frac),
                                           ctypes.c_uint64(int(seed) & (2 ** 64 - 1)), stream))
    return chars


__all__ = ["aa_array", "substitute", "normrows", "probdict", "augment_seq", "augment_packed"]
