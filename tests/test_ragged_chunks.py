"""The row-piece ("ragged") forms of the (B,P) chunk kernels -- k_tokenize_chunks<.., RG> for every element type and for
the channels-first one-hot (B,C,P), k_tokens_bp8<.., RG> for int8 -- which take any padlen and any element-aligned
output instead of falling back to the row / generic kernels.  Bit-exact against the oracle
(/root/reference/src/tokenize.h:454-479 tokens, :342-369 one-hot transposed as bioseq/loaders.py:74 does)."""
import ctypes

import numpy as np
import pytest

from test_tokens8 import nasty_batch

pytestmark = pytest.mark.gpu

NP = {"b": np.int8, "h": np.int16, "i": np.int32, "l": np.uint64, "f": np.float32, "d": np.float64}
TOKS = (("AMINO20", (0, 0, 0)), ("DNA", (1, 1, 1)), ("SEB8", (1, 0, 1)))


def run_tokens(lib, capi, desc, dch, dof, B, P, dc, gpu, shift_elems):
    import torch
    dt = ctypes.c_int(0)
    capi.check(lib.bsq_dtype_from_destchar(dc.encode(), ctypes.byref(dt)))
    sz = lib.bsq_dtype_size(dt)
    buf = torch.full(((B * P + 24) * sz,), 0x5A, dtype=torch.uint8, device=gpu)
    lo = shift_elems * sz
    capi.check(lib.bsq_tokenize_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), B, P, 1, dt,
                                       buf.data_ptr() + lo, None))
    torch.cuda.synchronize()
    host = buf.cpu().numpy()
    assert (host[:lo] == 0x5A).all() and (host[lo + B * P * sz:] == 0x5A).all(), "wrote outside the matrix"
    return host[lo:lo + B * P * sz]


@pytest.mark.parametrize("dc", list("bhilfd"))
@pytest.mark.parametrize("B,lo,hi,P", [(1, 0, 1, 3), (7, 0, 5, 7), (300, 0, 30, 33), (1000, 1, 120, 125), (513, 0, 248, 251),
                                       (4097, 0, 60, 62), (64, 900, 1000, 1003), (2000, 0, 14, 16)])
def test_tokens_any_padlen_any_alignment(gpu, oracle, dc, B, lo, hi, P):
    import torch
    from bioseq_amd import capi
    lib = capi.load()
    hi = min(hi, P - 2) if P > 2 else 0
    chars, offs = nasty_batch(B + 31 * P, B, min(lo, hi), hi)
    dch, dof = torch.from_numpy(np.concatenate([chars, np.zeros(1, np.uint8)])).to(gpu), torch.from_numpy(offs).to(gpu)
    for knob in (0, 1):  # int8: k_tokens_bp8 (padlen >= 128) or k_tokenize_chunks
        capi.check(lib.bsq_tuning_set(b"tokens8", knob))
        try:
            for key, flags in TOKS:
                if hi + flags[0] + flags[1] > P:
                    continue
                want = oracle.OracleTokenizer(key, *flags).tokenize_packed(chars, offs, P, dc, True)
                assert want.dtype == NP[dc]
                for shift in (0, 1, 3):
                    got = run_tokens(lib, capi, capi.make_desc(key, *flags), dch, dof, B, P, dc, gpu, shift)
                    assert got.tobytes() == want.tobytes(), (key, flags, dc, shift, knob)
        finally:
            capi.check(lib.bsq_tuning_set(b"tokens8", 0))
        if dc != "b":
            break


@pytest.mark.parametrize("dc", list("bhilfd"))
@pytest.mark.parametrize("B,lo,hi,P", [(5, 0, 5, 7), (200, 0, 30, 33), (300, 1, 120, 125), (129, 0, 248, 251), (40, 0, 14, 16)])
def test_channels_first_onehot_any_padlen_any_alignment(gpu, oracle, dc, B, lo, hi, P):
    import torch
    from bioseq_amd import capi
    lib = capi.load()
    hi = min(hi, P - 2)
    chars, offs = nasty_batch(B + 17 * P, B, min(lo, hi), hi)
    rng = np.random.default_rng(B)
    mask = (rng.random(chars.size) < 0.8).astype(np.uint8)
    dch, dof = torch.from_numpy(np.concatenate([chars, np.zeros(1, np.uint8)])).to(gpu), torch.from_numpy(offs).to(gpu)
    dm = torch.from_numpy(np.concatenate([mask, np.zeros(1, np.uint8)])).to(gpu)
    dt = ctypes.c_int(0)
    capi.check(lib.bsq_dtype_from_destchar(dc.encode(), ctypes.byref(dt)))
    sz = lib.bsq_dtype_size(dt)
    for key, flags in TOKS:
        ora = oracle.OracleTokenizer(key, *flags)
        C = ora.alphabet_size()
        for use_mask in (False, True):
            want = ora.onehot_packed(chars, offs, P, dc, mask=mask if use_mask else None)  # (P, B, C)
            want = np.ascontiguousarray(want.transpose(1, 2, 0))                            # (B, C, P)
            for shift in (0, 1, 2):
                n = B * C * P * sz
                buf = torch.full((n + 24 * sz,), 0x5A, dtype=torch.uint8, device=gpu)
                lo_b = shift * sz
                capi.check(lib.bsq_onehot_bcl_device(ctypes.byref(capi.make_desc(key, *flags)), dch.data_ptr(), dof.data_ptr(),
                                                     dm.data_ptr() if use_mask else None, B, P, dt, buf.data_ptr() + lo_b, None))
                torch.cuda.synchronize()
                host = buf.cpu().numpy()
                assert (host[:lo_b] == 0x5A).all() and (host[lo_b + n:] == 0x5A).all(), "wrote outside the tensor"
                assert host[lo_b:lo_b + n].tobytes() == want.tobytes(), (key, flags, dc, use_mask, shift)


@pytest.mark.parametrize("dc", list("bhilfd"))
def test_wide_index_arithmetic_gives_the_same_results(gpu, oracle, dc):
    """The 64-bit index paths of the (B,P) chunk kernels (more than 2^31 16-byte pieces: > 32 GB of output) forced with the
    knob `wide_index` on small shapes, aligned and ragged, tokens and channels-first one-hot."""
    import torch
    from bioseq_amd import capi
    lib = capi.load()
    dt = ctypes.c_int(0)
    capi.check(lib.bsq_dtype_from_destchar(dc.encode(), ctypes.byref(dt)))
    sz = lib.bsq_dtype_size(dt)
    capi.check(lib.bsq_tuning_set(b"wide_index", 1))
    try:
        for B, P in ((700, 144), (300, 250), (1000, 129), (64, 1024)):
            chars, offs = nasty_batch(B + P, B, 0, P - 2)
            dch = torch.from_numpy(np.concatenate([chars, np.zeros(1, np.uint8)])).to(gpu)
            dof = torch.from_numpy(offs).to(gpu)
            for key, flags in TOKS:
                ora = oracle.OracleTokenizer(key, *flags)
                desc = capi.make_desc(key, *flags)
                for knob in ((0, 1) if dc == "b" else (0,)):
                    capi.check(lib.bsq_tuning_set(b"tokens8", knob))
                    got = run_tokens(lib, capi, desc, dch, dof, B, P, dc, gpu, 0)
                    capi.check(lib.bsq_tuning_set(b"tokens8", 0))
                    assert got.tobytes() == ora.tokenize_packed(chars, offs, P, dc, True).tobytes(), (key, B, P, knob)
                C = ora.alphabet_size()
                out = torch.full((B * C * P * sz,), 0x5A, dtype=torch.uint8, device=gpu)
                for path in (1, 2):
                    capi.check(lib.bsq_tuning_set(b"bcl_path", path))
                    capi.check(lib.bsq_onehot_bcl_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), None, B, P, dt,
                                                         out.data_ptr(), None))
                    torch.cuda.synchronize()
                    want = np.ascontiguousarray(ora.onehot_packed(chars, offs, P, dc).transpose(1, 2, 0))
                    assert out.cpu().numpy().tobytes() == want.tobytes(), (key, B, P, "bcl", path)
    finally:
        for k in (b"wide_index", b"tokens8", b"bcl_path"):
            capi.check(lib.bsq_tuning_set(k, 0))
