"""k_tokens_bp8 -- the (B,P) int8 token matrix kernel of BASELINE cfg2 / cfg5 (bioseq_amd/csrc/bsq_tokens8.hip) --
against the oracle (/root/reference/src/tokenize.h:454-479 restated in oracle/bsq_oracle.c), for both of its
alphabet-lookup forms (register table through v_perm_b32, LDS byte table), and against the round-1 kernel it
replaced.  Bit-exact."""
import ctypes
import itertools

import numpy as np
import pytest

from bioseq_amd import synth

pytestmark = pytest.mark.gpu

ALL_KEYS = ["AMINO", "AMINO20", "C", "DAYHOFF", "DNA", "DNA4", "DNA5", "DNAMETH", "KETO", "LIA10",
            "LIB10", "MURPHY", "PROTEIN", "PURPYR", "SEB10", "SEB14", "SEB6", "SEB8", "SEV10"]
COMBOS = list(itertools.product([0, 1], repeat=3))  # (eos, bos, padchar)


def nasty_batch(seed, n, lo, hi):
    """Ragged batch in which every byte value appears: non-letters, digits, bytes >= 0x80, both cases."""
    lens = synth.synth_lengths(seed, n, lo, hi)
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    rng = np.random.default_rng(seed)
    chars = rng.integers(0, 256, size=int(offs[-1]), dtype=np.uint8)
    letters = np.frombuffer((synth.AA + synth.AA.lower()).encode(), dtype=np.uint8)
    pick = rng.random(chars.size) < 0.7   # mostly residues, so that whole waves also take the fast path
    chars[pick] = letters[rng.integers(0, letters.size, size=int(pick.sum()))]
    return chars, offs


@pytest.fixture(params=[(0, 0, 0), (0, 1, 0), (0, 2, 0), (0, 2, 1), (1, 0, 0)],
                ids=["auto", "lds-table", "register-table", "register-table-round2-form", "round1-kernel"])
def variant(request):
    """auto / register-table take k_tokens_bp8_fast (round 3) on aligned shapes; tokens8_fast = 1 keeps the round-2 form.  (The LDS-DMA
    pipeline variants of round 5 left the library with their kernel: csrc/labs/README.md.)"""
    from bioseq_amd import capi
    lib = capi.load()
    off, lookup, nofast = request.param
    capi.check(lib.bsq_tuning_set(b"tokens8", off))
    capi.check(lib.bsq_tuning_set(b"tokens8_lookup", lookup))
    capi.check(lib.bsq_tuning_set(b"tokens8_fast", nofast))
    yield request.param[:2]
    capi.check(lib.bsq_tuning_set(b"tokens8", 0))
    capi.check(lib.bsq_tuning_set(b"tokens8_lookup", 0))
    capi.check(lib.bsq_tuning_set(b"tokens8_fast", 0))


def dev_tokens(lib, capi, desc, chars, offs, P, gpu, out_shift=0):
    import torch
    B = len(offs) - 1
    # one spare byte keeps the buffer non-empty; the kernel must never read it
    dch = torch.from_numpy(np.concatenate([chars, np.full(1, 0x41, np.uint8)])).to(gpu)[:len(chars)]
    dof = torch.from_numpy(offs).to(gpu)
    buf = torch.full((B * P + 48,), 99, dtype=torch.int8, device=gpu)
    out = buf[out_shift:out_shift + B * P]
    capi.check(lib.bsq_tokenize_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), B, P, 1, capi.I8,
                                       out.data_ptr(), None))
    torch.cuda.synchronize()
    host = buf.cpu().numpy()
    assert (host[:out_shift] == 99).all() and (host[out_shift + B * P:] == 99).all(), "wrote outside the matrix"
    return host[out_shift:out_shift + B * P].reshape(B, P)


def test_all_keys_and_flags_vs_oracle(gpu, oracle, variant):
    """19 letter alphabets x 8 flag combos, ragged batches of every byte value, P = 128 (the smallest the kernel
    takes) and 272 (rows straddle the 1-KiB stores)."""
    from bioseq_amd import capi
    lib = capi.load()
    for P in (128, 272):
        chars, offs = nasty_batch(4242 + P, 301, 0, P - 2)
        for key in ALL_KEYS:
            for (eos, bos, pad) in COMBOS:
                ora = oracle.OracleTokenizer(key, eos, bos, pad)
                want = ora.tokenize_packed(chars, offs, P, "b", True)
                got = dev_tokens(lib, capi, capi.make_desc(key, eos, bos, pad), chars, offs, P, gpu)
                assert got.tobytes() == want.tobytes(), (key, eos, bos, pad, P)


@pytest.mark.parametrize("B,lo,hi,P", [(1, 0, 0, 128), (1, 126, 126, 128), (3, 5, 9, 144), (65, 0, 4100, 4112),
                                       (40000, 0, 126, 128), (777, 510, 510, 512), (5000, 1, 254, 256),
                                       (33, 1000, 2046, 2048), (4097, 0, 14, 16 * 9)])
def test_shapes_vs_oracle(gpu, oracle, variant, B, lo, hi, P):
    """Single sequences, empty sequences, full rows, rows longer than a chunk, many short rows per chunk, a last
    chunk that is cut by the end of the matrix."""
    from bioseq_amd import capi
    lib = capi.load()
    chars, offs = synth.synth_packed(B * 7 + P, B, lo, hi, synth.DIRTY)
    for key, flags in (("AMINO20", (0, 0, 0)), ("DNA", (1, 1, 1)), ("SEB8", (1, 0, 1)), ("DNA5", (0, 1, 0))):
        if hi + flags[0] + flags[1] > P:
            continue
        ora = oracle.OracleTokenizer(key, *flags)
        want = ora.tokenize_packed(chars, offs, P, "b", True)
        got = dev_tokens(lib, capi, capi.make_desc(key, *flags), chars, offs, P, gpu)
        assert got.tobytes() == want.tobytes(), (key, flags)


@pytest.mark.parametrize("B,lo,hi,P", [(1, 0, 126, 129), (3, 120, 128, 130), (700, 0, 140, 143), (5000, 1, 248, 250),
                                       (4097, 0, 999, 1001), (2000, 500, 998, 1000), (257, 0, 134, 136),
                                       (65, 3000, 4100, 4111), (9000, 0, 126, 128)])
def test_ragged_padlen_and_misaligned_outputs(gpu, oracle, variant, B, lo, hi, P):
    """padlen that is not a multiple of 16 (the kernel's row-piece form: unaligned 16-byte stores, partial last piece of
    every row in 8 / 4 / 2 / 1-byte stores) and outputs at every byte alignment; nothing outside the matrix is written."""
    from bioseq_amd import capi
    lib = capi.load()
    chars, offs = nasty_batch(B * 3 + P, B, lo, hi)
    for key, flags in (("AMINO20", (0, 0, 0)), ("DNA", (1, 1, 1)), ("SEB8", (1, 0, 1)), ("DNA5", (0, 1, 0))):
        if hi + flags[0] + flags[1] > P:
            continue
        want = oracle.OracleTokenizer(key, *flags).tokenize_packed(chars, offs, P, "b", True)
        for out_shift in (0, 1, 4, 8, 13, 16):
            if P % 16 == 0 and out_shift in (0, 16):
                continue  # the aligned form: covered above
            got = dev_tokens(lib, capi, capi.make_desc(key, *flags), chars, offs, P, gpu, out_shift)
            assert got.tobytes() == want.tobytes(), (key, flags, out_shift)


def test_tables_that_do_not_fold_use_the_lds_lookup(gpu, variant):
    """A caller-made bsq_desc whose table maps digits, or the two cases differently, cannot use the 32-entry folded
    register table: the kernel must fall back to the byte table and still equal the generic element kernel."""
    import torch
    from bioseq_amd import capi
    lib = capi.load()
    chars, offs = nasty_batch(31337, 2000, 0, 254)
    B, P = len(offs) - 1, 256
    dch, dof = torch.from_numpy(chars).to(gpu), torch.from_numpy(offs).to(gpu)
    for trial in range(3):
        d = capi.make_desc("AMINO20", 1, 1, 1)
        if trial == 0:
            d.lut[ord("7")] = 3           # a mapped non-letter
        elif trial == 1:
            d.lut[ord("a")] = 5           # 'a' and 'A' disagree
        else:
            d.lut[ord("q")] = -1          # one case unmapped
        a = torch.full((B, P), 99, dtype=torch.int8, device=gpu)
        b = torch.full((B, P), 98, dtype=torch.int8, device=gpu)
        capi.check(lib.bsq_tokenize_device(ctypes.byref(d), dch.data_ptr(), dof.data_ptr(), B, P, 1, capi.I8, a.data_ptr(), None))
        capi.check(lib.bsq_tokenize_device_generic(ctypes.byref(d), dch.data_ptr(), dof.data_ptr(), B, P, 1, capi.I8, b.data_ptr(), None))
        torch.cuda.synchronize()
        assert torch.equal(a, b), trial


def test_unvalidated_overlong_sequences_are_clamped(gpu, variant):
    """The device entry point does not validate: a sequence longer than the row must be clamped to it (memory safety),
    exactly as the generic kernel does."""
    import torch
    from bioseq_amd import capi
    lib = capi.load()
    chars, offs = synth.synth_packed(5, 300, 100, 400, synth.AA)
    B, P = len(offs) - 1, 128
    dch, dof = torch.from_numpy(chars).to(gpu), torch.from_numpy(offs).to(gpu)
    for flags in ((0, 0, 0), (1, 1, 1)):
        d = capi.make_desc("AMINO20", *flags)
        a = torch.full((B, P), 99, dtype=torch.int8, device=gpu)
        b = torch.full((B, P), 98, dtype=torch.int8, device=gpu)
        capi.check(lib.bsq_tokenize_device(ctypes.byref(d), dch.data_ptr(), dof.data_ptr(), B, P, 1, capi.I8, a.data_ptr(), None))
        capi.check(lib.bsq_tokenize_device_generic(ctypes.byref(d), dch.data_ptr(), dof.data_ptr(), B, P, 1, capi.I8, b.data_ptr(), None))
        torch.cuda.synchronize()
        assert torch.equal(a, b), flags


@pytest.mark.slow
def test_more_than_2_31_output_elements(gpu, bsq):
    """B * padlen >= 2^31: the kernel's chunk coordinates leave 32-bit arithmetic (div_by path).  Built and checked ON
    the device against plain torch indexing, both lookup forms."""
    import torch
    from bioseq_amd import capi
    lib = capi.load()
    B, lo, hi, P = 2_200_000, 900, 1000, 1008
    assert B * P >= (1 << 31) and P % 16 == 0
    g = torch.Generator(device=gpu).manual_seed(7)
    lens = torch.randint(lo, hi + 1, (B,), device=gpu, generator=g, dtype=torch.int64)
    offs = torch.zeros(B + 1, dtype=torch.int64, device=gpu)
    offs[1:] = torch.cumsum(lens, 0)
    total = int(offs[-1])
    letters = torch.tensor(list(b"ACGTNacgt*"), dtype=torch.uint8, device=gpu)   # N and * are unmapped in DNA4
    chars = letters[torch.randint(0, letters.numel(), (total,), device=gpu, generator=g)]
    tok = bsq.Tokenizer("DNA4", 1, 1, 1)
    lut = torch.from_numpy(np.asarray(tok.byte_table())).to(gpu).to(torch.int16)

    def expected(b0, b1):
        pos = torch.arange(P, device=gpu)[None, :] - 1
        L = lens[b0:b1, None]
        idx = (offs[b0:b1, None] + pos).clamp_(0, total - 1)
        ids = lut[chars[idx].long()]
        ids = torch.where(pos < 0, torch.full_like(ids, tok.bos()), ids)
        ids = torch.where(pos == L, torch.full_like(ids, tok.eos()), ids)
        ids = torch.where(pos > L, torch.full_like(ids, tok.pad()), ids)
        return ids.clamp_(min=0).to(torch.int8)

    for lookup in (2, 1):
        capi.check(lib.bsq_tuning_set(b"tokens8_lookup", lookup))
        try:
            t = tok.tokenize_packed(chars, offs, P, "B", True)
        finally:
            capi.check(lib.bsq_tuning_set(b"tokens8_lookup", 0))
        for b0 in list(range(0, B, 400_000)) + [B - 50_000]:
            b1 = min(B, b0 + 50_000)
            assert torch.equal(t[b0:b1], expected(b0, b1)), (lookup, b0)
        del t
