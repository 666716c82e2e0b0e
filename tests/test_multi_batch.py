"""bsq_tokenize_device_multi (round 6): n independent packed batches of one tokenizer / padlen / layout / element type in ONE launch
(k_tokens_bp8_fast_multi, k_tokens_pb8_fast_multi: the grid is the concatenation of the batches' grids, the per-batch pointers a table
in the kernel arguments) -- against the oracle (/root/reference/src/tokenize.h:381-485 restated in oracle/bsq_oracle.c) batch by
batch, bit-exact, for qualifying groups (one launch), mixed groups (fall back to n launches), empty batches, more than eight batches,
and with guard bytes around every output."""
import ctypes
import itertools

import numpy as np
import pytest

from bioseq_amd import synth

pytestmark = pytest.mark.gpu

DT = {"b": ("I8", np.int8), "h": ("I16", np.int16), "i": ("I32", np.int32), "f": ("F32", np.float32)}


def _batch(seed, n, lo, hi, nasty=False):
    lens = synth.synth_lengths(seed, n, lo, hi)
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    rng = np.random.default_rng(seed)
    letters = np.frombuffer((synth.AA + synth.AA.lower()).encode(), dtype=np.uint8)
    chars = letters[rng.integers(0, letters.size, size=int(offs[-1]))].copy()
    if nasty and chars.size:
        k = rng.random(chars.size) < 0.05
        chars[k] = rng.integers(0, 256, size=int(k.sum()), dtype=np.uint8)
    return chars, offs


def _run_multi(lib, capi, desc, batches, P, batch_first, destchar, gpu, stream=None):
    """batches: list of (chars, offs) numpy pairs -> list of numpy results; every output sits between 64 guard elements"""
    import torch
    code, npdt = DT[destchar]
    tdt = {np.int8: torch.int8, np.int16: torch.int16, np.int32: torch.int32, np.float32: torch.float32}[npdt]
    n = len(batches)
    arr = (capi.Batch * max(n, 1))()
    keep, outs = [], []
    for i, (chars, offs) in enumerate(batches):
        B = len(offs) - 1
        dch = torch.from_numpy(np.concatenate([chars, np.full(16, 0x41, np.uint8)])).to(gpu)
        dof = torch.from_numpy(offs).to(gpu)
        buf = torch.full((B * P + 128,), 99, dtype=tdt, device=gpu)
        keep.append((dch, dof, buf))
        outs.append((buf, B))
        arr[i].chars, arr[i].offsets, arr[i].B, arr[i].out = dch.data_ptr(), dof.data_ptr(), B, buf[64:].data_ptr()
    capi.check(lib.bsq_tokenize_device_multi(ctypes.byref(desc), n, arr, P, int(batch_first), getattr(capi, code), stream))
    torch.cuda.synchronize()
    res = []
    for buf, B in outs:
        h = buf.cpu().numpy()
        assert (h[:64] == 99).all() and (h[64 + B * P:] == 99).all(), "wrote outside a batch's matrix"
        res.append(h[64:64 + B * P].reshape((B, P) if batch_first else (P, B)))
    return res


def _want(oracle, key, flags, batches, P, batch_first, destchar):
    ora = oracle.OracleTokenizer(key, *flags)
    return [ora.tokenize_packed(c, o, P, destchar, batch_first) if len(o) > 1 else
            np.zeros((0, P) if batch_first else (P, 0), dtype=DT[destchar][1]) for c, o in batches]


@pytest.mark.parametrize("batch_first", [True, False], ids=["BP", "PB"])
def test_multi_equals_oracle_all_flags(gpu, oracle, batch_first):
    """2 .. 8 qualifying batches (sequence counts multiples of 64, so that the (P,B) rows are 64-byte aligned) of different sizes,
    every flag combination, three alphabets (foldable register table; DNA with BOS ids; a 250-class-free LDS table is covered below)"""
    from bioseq_amd import capi
    lib = capi.load()
    P = 192
    sizes = [64, 1024, 320, 2048, 128, 704, 4096, 256]
    batches = [_batch(100 + i, b, 0, P - 2, nasty=True) for i, b in enumerate(sizes)]
    for key in ("AMINO20", "DNA", "SEB8"):
        for flags in itertools.product([0, 1], repeat=3):
            desc = capi.make_desc(key, *flags)
            for n in (2, 3, 5, 8):
                got = _run_multi(lib, capi, desc, batches[:n], P, batch_first, "b", gpu)
                want = _want(oracle, key, flags, batches[:n], P, batch_first, "b")
                for i in range(n):
                    assert np.array_equal(got[i], want[i]), (key, flags, n, i)


@pytest.mark.parametrize("batch_first", [True, False], ids=["BP", "PB"])
def test_more_than_eight_batches_empty_ones_and_mixed_groups(gpu, oracle, batch_first):
    """19 batches (three launches of the fast form), some of them EMPTY (skipped), then a group in which one batch does not qualify
    (a sequence count that is no multiple of 64 / an odd count: the whole group runs as single launches) -- same results."""
    from bioseq_amd import capi
    lib = capi.load()
    P = 256
    key, flags = "PROTEIN", (1, 1, 1)
    desc = capi.make_desc(key, *flags)
    sizes = [128, 0, 64, 640, 0, 0, 192, 1280, 64, 64, 0, 2560, 448, 64, 832, 0, 128, 1920, 64]
    batches = [_batch(300 + i, b, 0, P - 2) for i, b in enumerate(sizes)]
    got = _run_multi(lib, capi, desc, batches, P, batch_first, "b", gpu)
    want = _want(oracle, key, flags, batches, P, batch_first, "b")
    for i in range(len(sizes)):
        assert got[i].shape == want[i].shape and np.array_equal(got[i], want[i]), i
    sizes = [128, 333, 1000, 64, 1]
    batches = [_batch(400 + i, b, 0, P - 2) for i, b in enumerate(sizes)]
    got = _run_multi(lib, capi, desc, batches, P, batch_first, "b", gpu)
    want = _want(oracle, key, flags, batches, P, batch_first, "b")
    for i in range(len(sizes)):
        assert np.array_equal(got[i], want[i]), i


def test_other_types_and_shapes_fall_back_or_run_multi(gpu, oracle):
    """int16 (P,B) runs the multi kernel; int32 / float32 and a padlen that is no multiple of 16 take the single launches; BYTES-like
    LDS-table alphabets ("lds-table" knob) the LDS form of the multi kernels.  All equal the oracle."""
    from bioseq_amd import capi
    lib = capi.load()
    key, flags = "AMINO", (1, 0, 1)
    desc = capi.make_desc(key, *flags)
    for P, destchar, bf in ((160, "h", False), (160, "h", True), (144, "i", False), (200, "b", True), (130, "b", False), (128, "f", True)):
        batches = [_batch(500 + i, b, 0, P - 2, nasty=True) for i, b in enumerate([256, 64, 1088])]
        got = _run_multi(lib, capi, desc, batches, P, bf, destchar, gpu)
        want = _want(oracle, key, flags, batches, P, bf, destchar)
        for i in range(3):
            assert np.array_equal(got[i], want[i]), (P, destchar, bf, i)
    capi.check(lib.bsq_tuning_set(b"tokens8_lookup", 1))
    try:
        for bf in (True, False):
            batches = [_batch(600 + i, b, 0, 190, nasty=True) for i, b in enumerate([256, 64, 1088])]
            got = _run_multi(lib, capi, desc, batches, 192, bf, "b", gpu)
            want = _want(oracle, key, flags, batches, 192, bf, "b")
            for i in range(3):
                assert np.array_equal(got[i], want[i]), (bf, i)
    finally:
        capi.check(lib.bsq_tuning_set(b"tokens8_lookup", 0))


def test_multi_equals_single_calls_at_cfg2_shape_and_on_a_side_stream(gpu):
    """Four batches of BASELINE config 2's shape (16 384 sequences each, padlen 1024) in one launch on a non-default stream == four
    bsq_tokenize_device calls, both layouts; and the argument errors."""
    import torch
    from bioseq_amd import capi
    lib = capi.load()
    c = synth.CONFIGS["cfg2"]
    desc = capi.make_desc(c["key"], c["eos"], c["bos"], c["padchar"])
    P, nb, B = c["padlen"], 4, 16384
    side = torch.cuda.Stream(device=gpu)
    sh = ctypes.c_void_p(side.cuda_stream)
    for bf in (1, 0):
        arr = (capi.Batch * nb)()
        keep = []
        for i in range(nb):
            ch, of = synth.synth_packed(c["seed"], B, c["lo"], c["hi"], c["letters"], first=i * B)
            dch, dof = torch.from_numpy(ch).to(gpu), torch.from_numpy(of).to(gpu)
            out = torch.full((B * P,), 77, dtype=torch.int8, device=gpu)
            ref = torch.full((B * P,), 78, dtype=torch.int8, device=gpu)
            capi.check(lib.bsq_tokenize_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), B, P, bf, capi.I8, ref.data_ptr(), None))
            keep.append((dch, dof, out, ref))
            arr[i].chars, arr[i].offsets, arr[i].B, arr[i].out = dch.data_ptr(), dof.data_ptr(), B, out.data_ptr()
        torch.cuda.synchronize()
        capi.check(lib.bsq_tokenize_device_multi(ctypes.byref(desc), nb, arr, P, bf, capi.I8, sh))
        side.synchronize()
        for i in range(nb):
            assert torch.equal(keep[i][2], keep[i][3]), (bf, i)
    assert lib.bsq_tokenize_device_multi(ctypes.byref(desc), -1, arr, P, 1, capi.I8, None) == capi.ERR_INVALID_ARG
    assert lib.bsq_tokenize_device_multi(ctypes.byref(desc), 2, None, P, 1, capi.I8, None) == capi.ERR_INVALID_ARG
    assert lib.bsq_tokenize_device_multi(ctypes.byref(desc), 0, None, P, 1, capi.I8, None) == capi.OK
    arr[1].out = None
    assert lib.bsq_tokenize_device_multi(ctypes.byref(desc), 2, arr, P, 1, capi.I8, None) == capi.ERR_INVALID_ARG


def test_augment_multi_equals_single_calls(gpu):
    """bsq_augment_device_multi / bsq_augment_tokenize_device_multi (one augmentation launch + one token launch per eight batches) ==
    the per-batch calls with the same seeds, bit for bit: mutated characters AND token matrices, 1 .. 11 batches incl. an empty one,
    both layouts; the mutation law itself is pinned on the single-batch entry by tests/test_augment.py."""
    import torch
    from bioseq_amd import capi
    lib = capi.load()
    key, flags = "SEB8", (0, 0, 1)
    desc = capi.make_desc(key, *flags)
    P = 256
    sizes = [512, 64, 0, 2048, 192, 1024, 64, 320, 128, 4096, 256]
    base = [_batch(900 + i, b, 1, P - 2) for i, b in enumerate(sizes)]
    seeds = [1000 + 7 * i for i in range(len(sizes))]
    for bf in (1, 0):
        for n in (1, 3, 8, 11):
            single_c, single_t, multi_c, multi_t = [], [], [], []
            arr = (capi.Batch * n)()
            sd = (ctypes.c_uint64 * n)(*seeds[:n])
            keep = []
            for i in range(n):
                chars, offs = base[i]
                B = len(offs) - 1
                dof = torch.from_numpy(offs).to(gpu)
                c1 = torch.from_numpy(np.concatenate([chars, np.full(16, 0x41, np.uint8)])).to(gpu)
                c2 = c1.clone()
                o1 = torch.full((max(B * P, 1),), 55, dtype=torch.int8, device=gpu)
                o2 = torch.full((max(B * P, 1),), 56, dtype=torch.int8, device=gpu)
                if B:
                    capi.check(lib.bsq_augment_tokenize_device(ctypes.byref(desc), c1.data_ptr(), dof.data_ptr(), B, P, bf, capi.I8, o1.data_ptr(), 1, 0.5,
                                                               seeds[i], None))
                arr[i].chars, arr[i].offsets, arr[i].B, arr[i].out = c2.data_ptr(), dof.data_ptr(), B, o2.data_ptr()
                keep.append((dof, c1, c2, o1, o2, B))
            capi.check(lib.bsq_augment_tokenize_device_multi(ctypes.byref(desc), n, arr, P, bf, capi.I8, 1, 0.5, sd, None))
            torch.cuda.synchronize()
            capi.check(lib.bsq_fused_status(None))
            for i, (dof, c1, c2, o1, o2, B) in enumerate(keep):
                assert torch.equal(c1, c2), ("mutated characters differ", bf, n, i)
                if B:
                    assert torch.equal(o1[:B * P], o2[:B * P]), ("tokens differ", bf, n, i)
                    changed = int((c2[:len(base[i][0])].cpu() != torch.from_numpy(base[i][0])).sum())
                    assert 0.3 * B < changed < 0.7 * B or B < 100
    # augmentation alone, chains of 3, every sequence (frac 1)
    n = 5
    arr = (capi.Batch * n)()
    sd = (ctypes.c_uint64 * n)(*seeds[:n])
    keep = []
    for i in range(n):
        chars, offs = base[i]
        B = len(offs) - 1
        dof = torch.from_numpy(offs).to(gpu)
        c1 = torch.from_numpy(np.concatenate([chars, np.full(16, 0x41, np.uint8)])).to(gpu)
        c2 = c1.clone()
        if B:
            capi.check(lib.bsq_augment_device(c1.data_ptr(), dof.data_ptr(), B, 3, 1.0, seeds[i], None))
        arr[i].chars, arr[i].offsets, arr[i].B, arr[i].out = c2.data_ptr(), dof.data_ptr(), B, None
        keep.append((dof, c1, c2))
    capi.check(lib.bsq_augment_device_multi(n, arr, 3, 1.0, sd, None))
    torch.cuda.synchronize()
    for dof, c1, c2 in keep:
        assert torch.equal(c1, c2)
    assert lib.bsq_augment_device_multi(2, arr, 1, 0.5, None, None) == capi.ERR_INVALID_ARG


def test_python_surface_of_the_multi_batch_calls(gpu, bsq, oracle):
    """bioseq_amd.multi.tokenize_packed_multi / augment_tokenize_packed_multi == the per-batch Python calls (and the oracle), on the caller's
    current stream; outs=; the reference's error for an over-long sequence."""
    import torch
    from bioseq_amd import blosum, multi
    tok, ora = bsq.Tokenizer("SEB8", 0, 1, 1), oracle.OracleTokenizer("SEB8", 0, 1, 1)
    P = 256
    host = [_batch(1300 + i, b, 1, P - 2) for i, b in enumerate([640, 64, 2048, 128, 960])]
    dev = [(torch.from_numpy(np.concatenate([c, np.full(16, 0x41, np.uint8)])).to(gpu)[:len(c)], torch.from_numpy(o).to(gpu)) for c, o in host]
    side = torch.cuda.Stream(device=gpu)
    for bf in (True, False):
        with torch.cuda.stream(side):
            got = multi.tokenize_packed_multi(tok, dev, P, "B", bf)
        side.synchronize()
        for (c, o), g in zip(host, got):
            assert g.cpu().numpy().tobytes() == ora.tokenize_packed(c, o, P, "B", bf).tobytes()
    outs = [torch.full((len(o) - 1, P), 9, dtype=torch.int8, device=gpu) for _, o in host]
    got = multi.tokenize_packed_multi(tok, dev, P, "b", True, outs=outs)
    assert all(a is b for a, b in zip(got, outs))
    # augmentation: the same seeds through the one-batch entry
    a1 = [(c.clone(), o) for c, o in dev]
    a2 = [(c.clone(), o) for c, o in dev]
    seeds = [11, 22, 33, 44, 55]
    want = [blosum.augment_tokenize_packed(tok, c, o, P, "b", True, chain_len=2, augment_frac=0.7, seed=s) for (c, o), s in zip(a1, seeds)]
    got = multi.augment_tokenize_packed_multi(tok, a2, P, "b", True, chain_len=2, augment_frac=0.7, seeds=seeds)
    torch.cuda.synchronize()
    for (c1, _), (c2, _), w, g in zip(a1, a2, want, got):
        assert torch.equal(c1, c2) and torch.equal(w, g)
    blosum.check_fused()
    bad_o = host[1][1].copy()
    bad_o[-1] += 0  # (lengths as they are) ...
    long_c = np.concatenate([host[1][0], np.full(300, 0x41, np.uint8)])
    long_o = host[1][1].copy()
    long_o[-1] += 300   # ... and one sequence that no longer fits
    with pytest.raises(RuntimeError, match="seq len \\+ bos \\+ eos > padlen"):
        multi.tokenize_packed_multi(tok, [dev[0], (torch.from_numpy(long_c).to(gpu), torch.from_numpy(long_o).to(gpu))], P, "B", True)


def test_a_batch_of_empty_sequences_among_the_batches(gpu, bsq, oracle):
    """a batch whose sequences are ALL empty has no characters: torch's data_ptr() of such a tensor is null, and the augmentation entry
    points refuse a null `chars` for B > 0 -- the Python surface must still take it (found by the randomised harness, round 6)"""
    import torch
    from bioseq_amd import multi
    tok, ora = bsq.Tokenizer("SEB8", 1, 1, 1), oracle.OracleTokenizer("SEB8", 1, 1, 1)
    P = 128
    c, o = _batch(77, 192, 1, P - 2)
    none_c, none_o = np.zeros(0, np.uint8), np.zeros(65, np.int64)
    host = [(c, o), (none_c, none_o), (c, o)]
    dev = [(torch.from_numpy(c_).to(gpu), torch.from_numpy(o_).to(gpu)) for c_, o_ in host]
    assert dev[1][0].data_ptr() == 0 or dev[1][0].numel() == 0
    for bf in (True, False):
        got = multi.tokenize_packed_multi(tok, dev, P, "b", bf)
        for (c_, o_), g in zip(host, got):
            assert g.cpu().numpy().tobytes() == ora.tokenize_packed(c_, o_, P, "b", bf).tobytes()
    work = [(c_.clone(), o_) for c_, o_ in dev]
    got = multi.augment_tokenize_packed_multi(tok, work, P, "b", True, chain_len=1, augment_frac=1.0, seeds=[1, 2, 3])
    torch.cuda.synchronize()
    assert got[1].cpu().numpy().tobytes() == ora.tokenize_packed(none_c, none_o, P, "b", True).tobytes()
    for k in (0, 2):   # one mutation per (non-empty) sequence, tokens of the mutated characters
        mutated = work[k][0].cpu().numpy()
        assert got[k].cpu().numpy().tobytes() == ora.tokenize_packed(mutated, o, P, "b", True).tobytes()
        assert 0 < int((mutated != c).sum()) <= 192
