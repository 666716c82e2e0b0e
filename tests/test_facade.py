"""the module-level facade `onehot_encode` / `f_encode` (reference bioseq/__init__.py:36-116)
against arrays the REFERENCE's own Python package produced (tests/golden/facade.npz, written by make_golden.py in the build
container from /root/reference/bioseq + oracle/_ref).  Every return form: numpy (the reference's default), torch tensor in
host memory (to_pytorch=True, device=None), and the tensor encoded straight on the HIP device (device='cuda')."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEQS = ["ACGT", "GGN", "", "acgtACGT"]


@pytest.fixture(scope="module")
def facade(golden_dir):
    return np.load(os.path.join(golden_dir, "facade.npz")), json.load(open(os.path.join(golden_dir, "facade.json")))


def same(got, want):
    assert got.shape == want.shape and got.dtype == want.dtype, (got.shape, got.dtype, want.shape, want.dtype)
    assert np.ascontiguousarray(got).tobytes() == np.ascontiguousarray(want).tobytes()


def test_f_encode_batch_every_return_form(gpu, bsq, facade):
    import torch
    A, J = facade
    want = A["f_encode_dna_bos_p10"]                        # (10, 4, 5) int8: destchar 'B' -> 'b' (tokenize.cpp:66)
    got = bsq.f_encode(SEQS, key="dna", bos=True, padlen=10)
    assert isinstance(got, np.ndarray)
    same(got, want)
    t = bsq.f_encode(SEQS, key="dna", bos=True, padlen=10, to_pytorch=True)
    assert isinstance(t, torch.Tensor) and not t.is_cuda
    same(t.numpy(), want)
    d = bsq.f_encode(SEQS, key="dna", bos=True, padlen=10, to_pytorch=True, device="cuda")
    assert d.is_cuda and d.dtype == torch.int8
    same(d.cpu().numpy(), want)
    d = bsq.f_encode(tuple(SEQS), key="dna", bos=True, padlen=10, to_pytorch=True, device=gpu)
    same(d.cpu().numpy(), want)

    want = A["f_encode_prot_pbeos_f_bf"]                    # (2, 13, 23) f32, batch_first
    kw = dict(key="PROTEIN", bos=True, eos=True, padchar=True, padlen=13, destchar="f", batch_first=True)
    same(bsq.f_encode(["MKV", "ACDEFGHIKL"], **kw), want)
    same(bsq.f_encode(["MKV", "ACDEFGHIKL"], to_pytorch=True, **kw).numpy(), want)
    d = bsq.f_encode(["MKV", "ACDEFGHIKL"], to_pytorch=True, device="cuda", **kw)
    assert d.is_cuda and tuple(d.shape) == want.shape and not d.is_contiguous()   # a strided view, like einops.rearrange
    same(d.cpu().numpy(), want)

    want = A["f_encode_torch_sf_i32"]                       # (6, 3, 7) int32, seq-first, torch return
    kw = dict(key="DNA5", eos=True, padchar=True, padlen=6, destchar="i", to_pytorch=True)
    t = bsq.f_encode(["ACGT", "NNNN", "acg"], **kw)
    assert str(t.dtype) == J["f_encode_torch_sf_i32"]["dtype"] and t.is_contiguous() == J["f_encode_torch_sf_i32"]["contiguous"]
    same(t.numpy(), want)
    d = bsq.f_encode(["ACGT", "NNNN", "acg"], device="cuda", **kw)
    assert d.is_cuda and str(d.dtype) == J["f_encode_torch_sf_i32"]["dtype"]
    same(d.cpu().numpy(), want)


def test_onehot_encode_batch_every_return_form(gpu, bsq, facade):
    import torch
    A, J = facade
    want = A["onehot_encode_torch_bf"]                      # (4, 11, 7) f32
    tok = bsq.pbeos_tokenizers["DNA"]
    kw = dict(padlen=11, destchar="f", batch_first=True)
    same(bsq.onehot_encode(tok, SEQS, **kw), want)
    t = bsq.onehot_encode(tok, SEQS, to_pytorch=True, **kw)
    assert str(t.dtype) == J["onehot_encode_torch_bf"]["dtype"] and t.is_contiguous() == J["onehot_encode_torch_bf"]["contiguous"]
    same(t.numpy(), want)
    d = bsq.onehot_encode(tok, SEQS, to_pytorch=True, device="cuda", **kw)
    assert d.is_cuda and d.is_contiguous() == J["onehot_encode_torch_bf"]["contiguous"]
    same(d.cpu().numpy(), want)
    # device given without to_pytorch: the reference returns the numpy array untouched (__init__.py:61)
    assert isinstance(bsq.onehot_encode(tok, SEQS, device="cuda", **kw), np.ndarray)

    want = A["onehot_encode_numpy_sf"]                      # (12, 3, 22) int16; str / bytes / bytearray items
    items = ["MKV", b"ACDEFGHIKL", bytearray(b"WY")]
    same(bsq.onehot_encode(bsq.beos_tokenizers["AMINO20"], items, padlen=12, destchar="h"), want)
    same(bsq.onehot_encode(bsq.beos_tokenizers["AMINO20"], items, padlen=12, destchar="h", to_pytorch=True,
                           device="cuda").cpu().numpy(), want)
    want = A["onehot_encode_numpy_bf"]                      # (3, 10, 9) f64, batch_first
    same(bsq.onehot_encode(bsq.pos_tokenizers["SEB8"], ["MKV", "ACDEFGHIKL", ""], padlen=10, destchar="d", batch_first=True), want)
    same(bsq.onehot_encode(bsq.pos_tokenizers["SEB8"], ["MKV", "ACDEFGHIKL", ""], padlen=10, destchar="d", batch_first=True,
                           to_pytorch=True, device="cuda").cpu().numpy(), want)


def test_single_sequence_through_the_facade(gpu, bsq, facade):
    import torch
    A, _ = facade
    same(bsq.f_encode("ACGT", key="DNA"), A["single_str_default"])
    same(bsq.pbeos_tokenizers["DNA"].onehot_encode("ACGT", 8, "f"), A["single_pbeos_p8"])
    same(bsq.DNATokenizer.onehot_encode(b"ACGTA"), A["single_bytes_default"])
    want = A["f_encode_single_torch"]                       # (11, 5) f32
    kw = dict(key="DNA", bos=True, padlen=10, destchar="f", to_pytorch=True)
    t = bsq.f_encode("ACGTTGCA", **kw)
    assert isinstance(t, torch.Tensor) and not t.is_cuda
    same(t.numpy(), want)
    d = bsq.f_encode("ACGTTGCA", device="cuda", **kw)
    assert isinstance(d, torch.Tensor) and d.is_cuda
    same(d.cpu().numpy(), want)
    d = bsq.onehot_encode(bsq.pbeos_tokenizers["DNA"], "ACGT", padlen=8, destchar="f", to_pytorch=True, device=gpu)
    assert d.is_cuda
    same(d.cpu().numpy(), A["single_pbeos_p8"])
