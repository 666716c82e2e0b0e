"""bioseq_amd -- MI355X-native batch tokenizer / one-hot encoder for DNA and protein sequences.

Drop-in for the tokenizer path of dnbaker/bioseq (`import bioseq_amd as bioseq`): the same
``Tokenizer(key, eos, bos, padchar)`` class with ``batch_tokenize`` / ``batch_onehot_encode``,
the pre-built tokenizer dictionaries and the ``onehot_encode`` / ``f_encode`` helpers of the
reference's ``bioseq/__init__.py:36-168`` -- with all encoding done by hand-written HIP kernels
for gfx950 behind the C ABI in ``include/bsq.h``.  There is no CPU fallback: without the native
extension (or without a HIP device) the encode calls raise.

Additions over the reference surface (all keyword-only / new names, defaults unchanged):

* ``batch_tokenize(..., device="cuda")`` / ``batch_onehot_encode(..., device="cuda")`` return a
  ``torch.Tensor`` that was produced on the device (no host round trip);
* ``Tokenizer.tokenize_packed`` / ``onehot_packed`` take an already packed batch
  (``chars uint8[total]``, ``offsets int64[B+1]``) as numpy arrays or device tensors;
* ``onehot_encode(..., device=...)`` / ``f_encode(..., device=...)`` encode straight on that device.
"""
from __future__ import annotations

from . import _hipruntime as _hipruntime

_hipruntime.preload()  # share torch's bundled HIP runtime when torch is installed (see _hipruntime.py)

try:
    from . import cbioseq
except ImportError as _e:  # fail loudly: the HIP extension IS the product
    raise ImportError(
        "bioseq_amd: the native extension is missing or failed to load (%s). Build it in-tree with "
        "`python bioseq_amd/build.py` (needs hipcc; gfx950 code objects cross-compile without a GPU)." % (_e,)
    ) from _e

from .cbioseq import Threading, Tokenizer, get_host_threads, get_num_threads, set_host_threads, set_num_threads  # noqa: F401
from . import synth  # noqa: F401
from . import blosum, multi, sharding  # noqa: F401
from .flatfile import FlatFile, FlatFileIterator, getstats  # noqa: F401  (bioseq.FlatFile / getstats, /root/reference/src/fxstats.cpp:166-219)

__version__ = "0.1.0"


def device_count() -> int:
    """Number of HIP devices the native library can see."""
    return cbioseq.device_count()


def _is_hip_device(device) -> bool:
    if device is None:
        return False
    import torch
    return torch.device(device).type == "cuda"


def onehot_encode(tokenizer, seqbatch, padlen=-1, destchar='B', batch_first=False, to_pytorch=False, device=None, *, devices=None):
    """One-hot encode a batch (or a single sequence) -- reference bioseq/__init__.py:36-66.

    seqbatch: list/tuple of str/bytes/bytearray -> ``batch_onehot_encode`` -> (padlen, B, C),
    or (B, padlen, C) as a strided view when ``batch_first``.  ``to_pytorch`` wraps the result in a
    tensor; with ``device`` set to a HIP device the batch is encoded directly on that device
    (the reference encodes on the host and copies, ``__init__.py:61-65``).

    ``devices=[...]`` (keyword-only, with ``to_pytorch``; not in the reference): the batch is sharded BY SEQUENCE over these HIP devices from
    this one process -- packed once on the host, every device receives and encodes its slice (``sharding.encode_on_devices``) -- and the
    list of per-device shards comes back, what a ``nn.DataParallel``-style consumer (training/cnnpretrain.py:85-94) feeds its replicas.
    """
    single = isinstance(seqbatch, (str, bytes))
    if devices is not None and (single or not to_pytorch):
        raise ValueError("devices= shards a BATCH over HIP devices and returns device tensors: pass a list of sequences and to_pytorch=True")
    if devices is not None:
        if padlen is None or padlen <= 0:
            raise ValueError("devices= needs an explicit padlen")
        shards = sharding.encode_on_devices(tokenizer, seqbatch, padlen, destchar, devices=devices, op="onehot")
        return [s.permute(1, 0, 2) for s in shards] if batch_first else shards
    on_device = to_pytorch and _is_hip_device(device)
    if single:  # one sequence: (max(L, padlen) + bos + eos, C), tokenize.h:188-216
        if on_device:
            encoded = tokenizer.onehot_encode(seqbatch, padlen, destchar, device=device)
        else:
            encoded = tokenizer.onehot_encode(seqbatch, padlen, destchar)
    else:
        encoded = tokenizer.batch_onehot_encode(seqbatch, padlen, destchar, device=device if on_device else None)
        if batch_first:  # 'seq batch base -> batch seq base', a strided view like the reference's einops.rearrange
            encoded = encoded.permute(1, 0, 2) if on_device else encoded.transpose(1, 0, 2)
    if on_device or not to_pytorch:
        return encoded
    import torch
    tensor = torch.from_numpy(encoded)
    return tensor if device is None else tensor.to(device)


def f_encode(seqbatch, key="DNA", bos=False, eos=False, padchar=False, padlen=-1, destchar='B', batch_first=False,
             to_pytorch=False, device=None, *, devices=None):
    """Functional form: build ``Tokenizer(key, bos=, eos=, padchar=)`` then ``onehot_encode``
    (reference bioseq/__init__.py:69-116)."""
    tokenizer = Tokenizer(key, bos=bos, eos=eos, padchar=padchar)
    return onehot_encode(tokenizer, seqbatch, padlen=padlen, destchar=destchar, batch_first=batch_first,
                         to_pytorch=to_pytorch, device=device, devices=devices)


# Pre-built tokenizers and dictionaries -- same names and keys as bioseq/__init__.py:119-156.
keys = ("SEB6", "SEB8", "SEB10", "SEV10", "MURPHY", "LIA10", "LIB10", "SEB6", "DAYHOFF", "DNA4", "DNA", "DNA5",
        "KETO", "PURPYR", "BYTES", "AMINO20", "PROTEIN")
bkeys = keys + tuple(map(str.lower, keys))

DNATokenizer = Tokenizer("DNA")
AmineTokenizer = Tokenizer("AMINO20")
Reduced6Tokenizer = Tokenizer("SEB6")
Reduced8Tokenizer = Tokenizer("SEB8")
Reduced10Tokenizer = Tokenizer("SEB10")
Reduced14Tokenizer = Tokenizer("SEB14")
DayhoffTokenizer = Tokenizer("DAYHOFF")
LIATokenizer = Tokenizer("LIA10")
LIBTokenizer = Tokenizer("LIB10")
# name -> shared plain tokenizer (no BOS / EOS / PAD); several names alias one alphabet, as in the reference
default_tokenizers = {}
for _names, _tok in ((("DNA",), DNATokenizer), (("AMINO20", "AMINE", "PROTEIN"), AmineTokenizer),
                     (("SEB6",), Reduced6Tokenizer), (("SEB8",), Reduced8Tokenizer), (("SEB10",), Reduced10Tokenizer),
                     (("SEB14",), Reduced14Tokenizer), (("LIA10", "LIA"), LIATokenizer), (("LIB10", "LIB"), LIBTokenizer)):
    for _name in _names:
        default_tokenizers[_name] = _tok
del _names, _tok, _name


def _family(bos, eos, padchar):
    return {k: Tokenizer(k, bos=bos, eos=eos, padchar=padchar) for k in bkeys}


pbeos_tokenizers = _family(True, True, True)
beos_tokenizers = _family(True, True, False)
pbos_tokenizers = _family(True, False, True)
bos_tokenizers = _family(True, False, False)
peos_tokenizers = _family(False, True, True)
eos_tokenizers = _family(False, True, False)
pos_tokenizers = _family(False, False, True)
total_tokenizer_dict = {(b, e, p, k): Tokenizer(k.upper(), bos=b, eos=e, padchar=p)
                        for b in (0, 1) for e in (0, 1) for p in (0, 1) for k in bkeys}


def get_tokenizer_dict(bos, eos, padchar):
    """Dictionary of tokenizers for a (bos, eos, padchar) combination (bioseq/__init__.py:159-168); all three off
    selects the short `default_tokenizers`."""
    table = {(1, 1, 1): pbeos_tokenizers, (1, 1, 0): beos_tokenizers, (1, 0, 1): pbos_tokenizers, (1, 0, 0): bos_tokenizers,
             (0, 1, 1): peos_tokenizers, (0, 1, 0): eos_tokenizers, (0, 0, 1): pos_tokenizers, (0, 0, 0): default_tokenizers}
    return table[(int(bool(bos)), int(bool(eos)), int(bool(padchar)))]


def make_embedding(tok, embdim, maxnorm=None, norm_type=2.0, scale_grad_by_freq=False, sparse=False, _weight=None):
    """``torch.nn.Embedding`` with one row per token of `tok` (PAD row as padding_idx when the tokenizer pads) --
    the helper of bioseq/__init__.py:171-188, same signature."""
    if norm_type < 1.0:
        raise AssertionError(f"{norm_type} is not >= 1., so it is not a norm.")
    from torch import nn
    pad_row = tok.pad() if tok.is_padded() else None
    return nn.Embedding(tok.alphabet_size(), embdim, padding_idx=pad_row, scale_grad_by_freq=scale_grad_by_freq,
                        sparse=sparse, _weight=_weight)


def torchify(arr):
    """numpy array -> torch tensor sharing its memory (bioseq/__init__.py:191-195)."""
    import torch
    return torch.from_numpy(arr)


def __getattr__(name):
    """`bioseq_amd.loaders` (FlatFileDataset, FF2NP, ...) imports torch: load it on first use only."""
    if name == "loaders":
        import importlib
        return importlib.import_module(".loaders", __name__)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))


__all__ = ["onehot_encode", "cbioseq", "f_encode", "Tokenizer", "make_embedding", "bos_tokenizers",
           "eos_tokenizers", "beos_tokenizers", "pbeos_tokenizers", "peos_tokenizers", "pbos_tokenizers",
           "pos_tokenizers", "default_tokenizers", "total_tokenizer_dict", "get_tokenizer_dict", "DNATokenizer",
           "AmineTokenizer", "Reduced6Tokenizer", "Reduced8Tokenizer", "Reduced10Tokenizer", "Reduced14Tokenizer",
           "DayhoffTokenizer", "LIATokenizer", "LIBTokenizer", "torchify", "set_num_threads", "get_num_threads",
           "Threading", "device_count", "synth", "blosum", "multi", "sharding", "FlatFile", "FlatFileIterator", "getstats", "loaders", "set_host_threads", "get_host_threads"]
