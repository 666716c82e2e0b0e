"""BLOSUM62 augmentation.  CPU: the transition table is bit-identical to the
reference's.  GPU: the kernel equals a numpy twin of its counter-based algorithm bit for bit, keeps the
reference's invariants (<= chain_len positions change, new != old, unknown residues use the X row,
augment_frac) and reproduces normrows statistically (chi-square)."""
import hashlib
import os

import ctypes

import numpy as np
import pytest

M64 = (1 << 64) - 1
LETTERS = "ARNDCQEGHILKMFPSTWYV"


def test_normrows_bit_exact(golden_dir):
    from bioseq_amd import blosum
    ref = np.load(os.path.join(golden_dir, "blosum_normrows.npy"))
    assert blosum.normrows.shape == (21, 20) and blosum.normrows.tobytes() == ref.tobytes()
    assert hashlib.sha256(blosum.normrows.astype("<f8").tobytes()).hexdigest() == \
        "8d08113767fb5ce5e059295759de0aad90b2c11eeb16fb14052e6ed5211436e9"
    assert np.allclose(blosum.normrows.sum(axis=1), 1.0)
    assert "".join(blosum.aa_array) == LETTERS
    assert blosum.probdict["H"].argmax() == LETTERS.index("H")  # the reference's import-time sanity check
    assert blosum.probdict["K"].argmax() == LETTERS.index("K")


def test_host_helpers():
    from bioseq_amd import blosum
    s = blosum.substitute("H", size=2000)
    vals, counts = np.unique(s, return_counts=True)
    assert vals[counts.argmax()] == "H"
    a = blosum.augment_seq("ACDEFGHIKLMNPQRSTVWY", 3)
    assert len(a) == 20 and 1 <= sum(x != y for x, y in zip(a, "ACDEFGHIKLMNPQRSTVWY")) <= 3


def _mix(z):
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def _rnd(seed, seq, i):
    return _mix((_mix((seed + 0x9E3779B97F4A7C15 * (seq + 1)) & M64) + 0xD1342543DE82EF95 * (i + 1)) & M64)


def twin(chars, offs, chain_len, frac, seed, normrows):
    """Pure-Python restatement of k_augment (bioseq_amd/csrc/bsq_augment.hip)."""
    out = chars.copy()
    cdf = np.cumsum(normrows, axis=1)  # left-to-right running sums, as the library builds them
    row_of = np.full(256, 20, dtype=np.int64)
    for c, ch in enumerate(LETTERS):
        row_of[ord(ch)] = c
    for b in range(len(offs) - 1):
        start, L = int(offs[b]), int(offs[b + 1] - offs[b])
        if L <= 0:
            continue
        if frac < 1.0 and not ((_rnd(seed, b, 0) >> 11) * 2.0 ** -53 < frac):
            continue
        ctr = 1
        for _ in range(chain_len):
            for _a in range(1 << 14):
                r = _rnd(seed, b, ctr)
                ctr += 1
                idx = (r * L) >> 64
                old = out[start + idx]
                row = int(row_of[old])
                pself = float(normrows[row, row]) if row < 20 else 0.0
                if (r & 0xFFFFFFFF) * 2.0 ** -32 < 1.0 - pself:  # position accepted
                    u = (_rnd(seed, b, ctr) >> 11) * 2.0 ** -53 * (cdf[row][19] - pself)
                    ctr += 1
                    pick = -1
                    for k in range(20):
                        if k == row:
                            continue
                        pick = k
                        if u < cdf[row][k] - (pself if k > row else 0.0):
                            break
                    out[start + idx] = ord(LETTERS[pick])
                    break
    return out


@pytest.mark.gpu
def test_kernel_equals_its_twin_and_invariants(gpu):
    import torch
    from bioseq_amd import blosum, synth
    chars, offs = synth.synth_packed(31, 400, 0, 60, synth.AA + "XBZxa*")
    from bioseq_amd import capi
    lib = capi.load()
    for chain_len, frac, seed in ((1, 1.0, 7), (3, 0.5, 123456789012345), (2, 1.5, 2 ** 63 + 5)):
        exp = twin(chars, offs, chain_len, frac, seed, blosum.normrows)
        for k in (1, 2, 4, 0):   # attempts per lane and round (knob augment_k): a speed matter only
            capi.check(lib.bsq_tuning_set(b"augment_k", k))
            try:
                d = blosum.augment_packed(torch.from_numpy(chars).to(gpu), torch.from_numpy(offs).to(gpu), chain_len, frac, seed)
            finally:
                capi.check(lib.bsq_tuning_set(b"augment_k", 0))
            got = d.cpu().numpy()
            assert got.tobytes() == exp.tobytes(), (chain_len, frac, seed, k)
        ndiff, touched = 0, 0
        for b in range(len(offs) - 1):
            a, g = chars[offs[b]:offs[b + 1]], got[offs[b]:offs[b + 1]]
            diff = np.nonzero(a != g)[0]
            assert len(diff) <= chain_len
            assert all(chr(g[i]) in LETTERS for i in diff)
            ndiff += len(diff)
            touched += len(diff) > 0
        nonempty = int((np.diff(offs) > 0).sum())
        if frac >= 1.0:
            assert touched >= nonempty * 0.9  # a chain can revert itself only for chain_len > 1
        else:
            assert 0.35 * nonempty < touched < 0.65 * nonempty
        # same seed -> same result; other seed -> different
        d2 = blosum.augment_packed(torch.from_numpy(chars).to(gpu), torch.from_numpy(offs).to(gpu), chain_len, frac, seed)
        assert torch.equal(d, d2)
    d3 = blosum.augment_packed(torch.from_numpy(chars).to(gpu), torch.from_numpy(offs).to(gpu), 1, 1.0, 8)
    assert not torch.equal(d3.cpu(), torch.from_numpy(twin(chars, offs, 1, 1.0, 7, blosum.normrows)))


@pytest.mark.gpu
def test_substitution_statistics_match_normrows(gpu):
    """chi-square of the empirical substitutions of 'H', 'W' and an unknown residue against normrows."""
    import torch
    from bioseq_amd import blosum
    n, L = 200000, 8
    for ch, row in (("H", LETTERS.index("H")), ("W", LETTERS.index("W")), ("x", 20)):
        chars = np.full(n * L, ord(ch), dtype=np.uint8)
        offs = np.arange(0, n * L + 1, L, dtype=np.int64)
        got = blosum.augment_packed(torch.from_numpy(chars).to(gpu), torch.from_numpy(offs).to(gpu), 1, 1.0, 99).cpu().numpy()
        changed = got[got != ord(ch)]
        assert changed.size == n  # exactly one substitution per sequence, never the same residue
        p = blosum.normrows[row].copy()
        if row < 20:
            p[row] = 0.0  # conditioned on new != old
        p /= p.sum()
        counts = np.array([(changed == ord(c)).sum() for c in LETTERS], dtype=np.float64)
        mask = p > 0
        chi2 = (((counts - n * p) ** 2)[mask] / (n * p)[mask]).sum()
        assert counts[~mask].sum() == 0
        assert chi2 < 60.0, (ch, chi2)  # 19 (18) dof: P(chi2 > 60) < 1e-5
        # positions are uniform over the sequence
        pos = np.nonzero(got.reshape(n, L) != ord(ch))[1]
        pc = np.bincount(pos, minlength=L).astype(np.float64)
        assert (((pc - n / L) ** 2) / (n / L)).sum() < 40.0


@pytest.mark.gpu
def test_cfg5_pipeline_augment_then_tokenize(gpu, bsq, oracle):
    """BASELINE config 5 at reduced batch: SEB8 tokens of the device-augmented batch equal the CPU
    encode of the same (post-augmentation) strings -- 'bit-exact on identical inputs'."""
    import torch
    from bioseq_amd import blosum, synth
    c = synth.CONFIGS["cfg5"]
    chars, offs = synth.synth_packed(c["seed"], 4096, c["lo"], c["hi"], c["letters"])
    dch, dof = torch.from_numpy(chars).to(gpu), torch.from_numpy(offs).to(gpu)
    blosum.augment_packed(dch, dof, chain_len=1, augment_frac=0.5, seed=505)  # AugmentedSeqDataset defaults
    mutated = dch.cpu().numpy()
    assert 0.4 * 4096 < int((np.add.reduceat((mutated != chars).astype(np.int64), offs[:-1]) > 0).sum()) < 0.6 * 4096
    tok = bsq.Tokenizer("SEB8")
    got = tok.tokenize_packed(dch, dof, c["padlen"], "B", True).cpu().numpy()
    exp = oracle.OracleTokenizer("SEB8").tokenize_packed(mutated, offs, c["padlen"], "B", True)
    assert got.tobytes() == exp.tobytes()


@pytest.mark.gpu
@pytest.mark.slow
@pytest.mark.parametrize("chain_len,frac", [(1, 0.5), (3, 1.0)])
def test_cfg5_full_size_augment_then_tokenize(gpu, bsq, oracle, chain_len, frac):
    """BASELINE config 5 AT ITS STATED SIZE (262 144 x 512, SEB8): BLOSUM62 augmentation in place on the device,
    then the token matrix -- tokens == the oracle's encode of the mutated bytes (bit-exact on identical inputs),
    at most chain_len residues change per sequence, lengths never change, about `frac` of the sequences are touched,
    and every changed residue became a DIFFERENT one of the 20 amino acids."""
    import torch
    from bioseq_amd import blosum, synth
    c = synth.CONFIGS["cfg5"]
    n, P = c["n"], c["padlen"]
    assert (n, P) == (262144, 512)
    chars, offs = synth.synth_packed(c["seed"], n, c["lo"], c["hi"], c["letters"])
    dch, dof = torch.from_numpy(chars).to(gpu), torch.from_numpy(offs).to(gpu)
    blosum.augment_packed(dch, dof, chain_len=chain_len, augment_frac=frac, seed=20260505 + chain_len)
    mutated = dch.cpu().numpy()
    diff = mutated != chars
    per_seq = np.add.reduceat(diff.astype(np.int64), offs[:-1])
    assert per_seq.max() <= chain_len and mutated.size == chars.size
    touched = int((per_seq > 0).sum())
    if frac >= 1.0:
        assert touched > 0.97 * n        # chains of 3 can undo themselves only rarely
    else:
        assert abs(touched - frac * n) < 6 * np.sqrt(n * frac * (1 - frac))   # binomial, 6 sigma
    aa = np.frombuffer(synth.AA.encode(), dtype=np.uint8)
    assert np.isin(mutated[diff], aa).all()
    tok = bsq.Tokenizer(c["key"], c["eos"], c["bos"], c["padchar"])
    got = tok.tokenize_packed(dch, dof, P, "B", True).cpu().numpy()
    exp = oracle.OracleTokenizer(c["key"], c["eos"], c["bos"], c["padchar"]).tokenize_packed(mutated, offs, P, "B", True, 8)
    assert got.tobytes() == exp.tobytes()


# ---------------------------------------------------------------- the joint law on a MIXED sequence

def _law_fixture(golden_dir):
    import json
    with open(os.path.join(golden_dir, "augment_law.json")) as f:
        J = json.load(f)
    assert J["letters"] == LETTERS and J["chain_len"] == 1
    return J["seq"], J["n"], np.array(J["counts"], dtype=np.float64)


def _expected_law(seq, normrows):
    """P(position i, new residue k) of the reference's reject-until-changed loop (blosum.py:75-83): every iteration
    draws i uniformly and k from normrows[r_i]; it stops when k != r_i.  So P(i, k) = normrows[r_i, k] / Z for
    k != r_i, Z = sum_i (1 - normrows[r_i, r_i]): the POSITION is proportional to 1 - p_self(residue)."""
    rows = [LETTERS.index(c) for c in seq]
    P = np.array([normrows[r] for r in rows])
    for i, r in enumerate(rows):
        P[i, r] = 0.0
    return P / P.sum()


def _chi2_p(counts, probs):
    from scipy import stats
    n = counts.sum()
    exp = n * probs
    keep = exp > 0
    assert counts[~keep].sum() == 0
    small = keep & (exp < 5)                      # pool thin cells
    c = np.append(counts[keep & ~small], counts[small].sum())
    e = np.append(exp[keep & ~small], exp[small].sum())
    if e[-1] == 0:
        c, e = c[:-1], e[:-1]
    return stats.chi2.sf((((c - e) ** 2) / e).sum(), len(c) - 1)


def test_reference_run_follows_the_stated_law(golden_dir):
    """CPU: the fixture (200 000 runs of the REFERENCE's augment_seq) against the closed form the kernel implements --
    position ~ (1 - p_self), new residue ~ row without its own entry.  Pins our reading of blosum.py:63-87."""
    from bioseq_amd import blosum
    seq, n, counts = _law_fixture(golden_dir)
    assert counts.sum() == n
    law = _expected_law(seq, blosum.normrows)
    assert _chi2_p(counts.sum(axis=1), law.sum(axis=1)) > 1e-6            # positions
    assert _chi2_p(counts.ravel(), law.ravel()) > 1e-6                    # joint
    w, a = seq.index("W"), seq.index("A")
    assert counts[w].sum() * 40 < counts[a].sum()                         # W (p_self 0.99) almost never, A often


def _kernel_table(gpu, seq, n, seed, chain_len=1):
    import torch
    from bioseq_amd import blosum
    L = len(seq)
    chars = np.tile(np.frombuffer(seq.encode(), dtype=np.uint8), n)
    offs = np.arange(0, n * L + 1, L, dtype=np.int64)
    got = blosum.augment_packed(torch.from_numpy(chars).to(gpu), torch.from_numpy(offs).to(gpu), chain_len, 1.0, seed).cpu().numpy()
    diff = (got != chars).reshape(n, L)
    return got.reshape(n, L), diff


@pytest.mark.gpu
def test_position_law_on_a_mixed_sequence(gpu, golden_dir):
    """200 000 copies of a fixed heterogeneous sequence, chain 1, frac 1: the histogram of the MUTATED POSITION against
    (1 - normrows[r, r]) / sum, per position the new residue against the row without its own entry, and a two-sample
    chi-square between the kernel's (position x residue) table and the reference's own run (augment_law.json)."""
    from scipy import stats
    from bioseq_amd import blosum
    seq, n, ref_counts = _law_fixture(golden_dir)
    got, diff = _kernel_table(gpu, seq, n, seed=20260303)
    assert (diff.sum(axis=1) == 1).all()                    # exactly one residue changes, never to itself
    pos = diff.argmax(axis=1)
    new = got[np.arange(n), pos]
    lut = np.full(256, -1)
    for k, ch in enumerate(LETTERS):
        lut[ord(ch)] = k
    assert (lut[new] >= 0).all()
    table = np.zeros((len(seq), 20))
    np.add.at(table, (pos, lut[new]), 1)
    law = _expected_law(seq, blosum.normrows)
    p_pos = _chi2_p(table.sum(axis=1), law.sum(axis=1))
    assert p_pos > 1e-6, ("position histogram", table.sum(axis=1), n * law.sum(axis=1))
    for i in range(len(seq)):
        if table[i].sum() >= 1000:
            assert _chi2_p(table[i], law[i] / law[i].sum()) > 1e-6, (i, seq[i])
    assert _chi2_p(table.ravel(), law.ravel()) > 1e-6
    # two-sample test against the reference's run (equal sample sizes): sum (k1 - k2)^2 / (k1 + k2)
    k1, k2 = table.ravel(), ref_counts.ravel()
    thin = (k1 + k2) < 20
    a = np.append(k1[~thin], k1[thin].sum())
    b = np.append(k2[~thin], k2[thin].sum())
    stat = (((a - b) ** 2) / (a + b)).sum()
    assert stats.chi2.sf(stat, len(a) - 1) > 1e-6, stat
    # an unknown residue (X row: p_self = 0) and lower case in the mix: accepted at once wherever they are drawn
    seq2 = "WWWWxWWWWWWa"
    got2, diff2 = _kernel_table(gpu, seq2, 50000, seed=5)
    pos2 = diff2.argmax(axis=1)
    law2 = np.array([1.0 - (blosum.normrows[17, 17] if c == "W" else 0.0) for c in seq2])
    assert _chi2_p(np.bincount(pos2, minlength=len(seq2)).astype(np.float64), law2 / law2.sum()) > 1e-6


@pytest.mark.gpu
def test_chain_of_mutations_follows_the_law_step_by_step(gpu):
    """chain_len 2: the second mutation sees the sequence the first one left (the weights move with it).  Checked through
    the marginal the closed form gives for the number of changed positions: P(both mutations hit the same position)."""
    from bioseq_amd import blosum
    seq, n = "AWHKCLGPSTYV", 100000
    got, diff = _kernel_table(gpu, seq, n, seed=77, chain_len=2)
    nd = diff.sum(axis=1)
    assert nd.max() <= 2
    law1 = _expected_law(seq, blosum.normrows)
    # P(second hits the same position i | first was (i, k)) = (1 - p_self(k)) / Z', Z' = Z - (1 - p_self(r_i)) + (1 - p_self(k))
    rows = [LETTERS.index(c) for c in seq]
    w = np.array([1.0 - blosum.normrows[r, r] for r in rows])
    p_same = 0.0
    for i in range(len(seq)):
        for k in range(20):
            if law1[i, k] > 0:
                wk = 1.0 - blosum.normrows[k, k]
                p_same += law1[i, k] * wk / (w.sum() - w[i] + wk)
    same = float((nd <= 1).sum())                # same position twice: one visible change (or none: mutated back)
    sd = np.sqrt(n * p_same * (1 - p_same))
    assert abs(same - n * p_same) < 6 * sd, (same, n * p_same)


@pytest.mark.gpu
@pytest.mark.parametrize("key,flags", [("SEB8", (0, 0, 0)), ("AMINO20", (1, 1, 1)), ("SEB14", (1, 0, 0)), ("BYTES", (0, 0, 0))])
@pytest.mark.parametrize("B,P,lo,hi", [(1, 128, 5, 100), (300, 128, 0, 126), (4096, 512, 50, 510), (20000, 256, 1, 254),
                                       (5000, 250, 0, 248), (70000, 160, 100, 158), (777, 1024, 900, 1022)])
def test_fused_augment_tokenize_equals_the_two_calls(gpu, bsq, oracle, key, flags, B, P, lo, hi):
    """bsq_augment_tokenize_device (ONE launch for (B,P) int8 where the fast token kernel applies: augmentation workgroups
    ahead of token workgroups that patch the mutated positions from the launch's side list once their rows' flags are up) ==
    bsq_augment_device, then bsq_tokenize_device: the same mutated characters, the same tokens (== the oracle's encode of the
    mutated bytes), for chains and fractions, layouts and types that fuse and that do not, and with the fusion switched off;
    no token wave ever gave up waiting."""
    import torch
    from bioseq_amd import blosum, capi, synth
    lib = capi.load()
    eos, bos, pad = flags
    hi = min(hi, P - eos - bos)
    lo = min(lo, hi)
    chars, offs = synth.synth_packed(B * 7 + P, B, lo, hi, synth.AA)
    dof = torch.from_numpy(offs).to(gpu)
    tok = bsq.Tokenizer(key, eos, bos, pad)
    ora = oracle.OracleTokenizer(key, eos, bos, pad)
    # knob augment_fused: 0 automatic (no-wait form + patch launch up to 16 384 chunks, else the flag form), 4 no-wait at any size, 1 two launches,
    # 2 / 3 the flag forms of rounds 3-4 (any XCD / same XCD)
    for chain_len, frac, dc, bf, knob in ((1, 0.5, "b", True, 0), (3, 1.0, "b", True, 0), (2, 0.7, "b", True, 1), (1, 0.5, "b", False, 0),
                                          (1, 1.0, "i", True, 0), (0, 1.0, "b", True, 0), (1, 0.5, "b", True, 2), (3, 1.0, "b", True, 2),
                                          (2, 0.7, "b", True, 3), (4, 1.0, "b", True, 4), (1, 0.5, "b", True, 4)):
        if key == "BYTES" and dc == "b":
            continue  # ids up to 255 need a wider type
        capi.check(lib.bsq_tuning_set(b"augment_fused", knob))
        try:
            ref = torch.from_numpy(chars).to(gpu)
            blosum.augment_packed(ref, dof, chain_len=chain_len, augment_frac=frac, seed=99 + chain_len)
            want_chars = ref.cpu().numpy()
            want = ora.tokenize_packed(want_chars, offs, P, dc, bf)
            got_chars = torch.from_numpy(chars).to(gpu)
            got = blosum.augment_tokenize_packed(tok, got_chars, dof, P, dc, bf, chain_len=chain_len, augment_frac=frac, seed=99 + chain_len)
            assert got_chars.cpu().numpy().tobytes() == want_chars.tobytes(), (key, flags, B, P, chain_len, frac, dc, bf, knob)
            assert got.cpu().numpy().tobytes() == want.tobytes(), (key, flags, B, P, chain_len, frac, dc, bf, knob)
        finally:
            capi.check(lib.bsq_tuning_set(b"augment_fused", 0))
    torch.cuda.synchronize()
    capi.check(lib.bsq_fused_status(None))


@pytest.mark.gpu
def test_fused_augment_tokenize_many_launches_two_streams(gpu, bsq, oracle):
    """Back-to-back fused launches on two streams (each has its own flag words and epochs), re-using the buffers: every
    result equals the two-call form."""
    import torch
    from bioseq_amd import blosum, capi, synth
    lib = capi.load()
    B, P = 30000, 256
    chars, offs = synth.synth_packed(4242, B, 10, 254, synth.AA)
    dof = torch.from_numpy(offs).to(gpu)
    tok = bsq.Tokenizer("SEB8")
    ora = oracle.OracleTokenizer("SEB8")
    streams = [torch.cuda.Stream(device=gpu), torch.cuda.Stream(device=gpu)]
    bufs = [torch.from_numpy(chars).to(gpu) for _ in streams]
    outs = [None, None]
    torch.cuda.synchronize()
    for it in range(12):
        for k, st in enumerate(streams):
            with torch.cuda.stream(st):
                outs[k] = blosum.augment_tokenize_packed(tok, bufs[k], dof, P, "b", True, chain_len=1, augment_frac=0.5, seed=1000 * k + it)
    torch.cuda.synchronize()
    for k in range(2):
        ref = torch.from_numpy(chars).to(gpu)
        for it in range(12):
            blosum.augment_packed(ref, dof, chain_len=1, augment_frac=0.5, seed=1000 * k + it)
        want_chars = ref.cpu().numpy()
        assert bufs[k].cpu().numpy().tobytes() == want_chars.tobytes(), k
        assert outs[k].cpu().numpy().tobytes() == ora.tokenize_packed(want_chars, offs, P, "b", True).tobytes(), k
    torch.cuda.synchronize()
    capi.check(lib.bsq_fused_status(None))


@pytest.mark.gpu
def test_fused_augment_tokenize_more_streams_than_flag_slots(gpu, bsq, oracle):
    """20 streams: the fused launch keeps flag words + side list for 16 (device, stream) pairs; the 17th evicts the least recently
    used one (round 3: it silently ran two launches forever) -- same results."""
    import torch
    from bioseq_amd import blosum, synth
    B, P = 2000, 128
    chars, offs = synth.synth_packed(77, B, 5, 126, synth.AA)
    dof = torch.from_numpy(offs).to(gpu)
    tok = bsq.Tokenizer("SEB8")
    ora = oracle.OracleTokenizer("SEB8")
    ref = torch.from_numpy(chars).to(gpu)
    blosum.augment_packed(ref, dof, chain_len=1, augment_frac=0.5, seed=5)
    want_chars = ref.cpu().numpy()
    want = ora.tokenize_packed(want_chars, offs, P, "b", True)
    streams = [torch.cuda.Stream(device=gpu) for _ in range(20)]
    res = []
    for st in streams:
        with torch.cuda.stream(st):
            buf = torch.from_numpy(chars).to(gpu)
            res.append((buf, blosum.augment_tokenize_packed(tok, buf, dof, P, "b", True, chain_len=1, augment_frac=0.5, seed=5)))
    torch.cuda.synchronize()
    for buf, out in res:
        assert buf.cpu().numpy().tobytes() == want_chars.tobytes() and out.cpu().numpy().tobytes() == want.tobytes()


@pytest.mark.gpu
def test_fused_entry_under_graph_capture_runs_the_two_launches(gpu, bsq, oracle):
    """A stream under capture cannot take the fused launch (a replayed graph would replay its epoch): the entry point records
    the two launches instead; capture, two replays == the two calls applied twice."""
    import torch
    from bioseq_amd import blosum, synth
    B, P = 5000, 256
    chars, offs = synth.synth_packed(31, B, 10, 254, synth.AA)
    dof = torch.from_numpy(offs).to(gpu)
    tok = bsq.Tokenizer("SEB8")
    ora = oracle.OracleTokenizer("SEB8")
    buf = torch.from_numpy(chars).to(gpu)
    out = torch.empty((B, P), dtype=torch.int8, device=gpu)
    side = torch.cuda.Stream(device=gpu)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):  # warm-up outside capture (tables, flag words)
        blosum.augment_tokenize_packed(tok, torch.from_numpy(chars).to(gpu), dof, P, "b", True, chain_len=1, augment_frac=0.5, seed=8, out=out)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        blosum.augment_tokenize_packed(tok, buf, dof, P, "b", True, chain_len=1, augment_frac=0.5, seed=8, out=out)
    buf.copy_(torch.from_numpy(chars).to(gpu))
    g.replay()
    g.replay()
    torch.cuda.synchronize()
    ref = torch.from_numpy(chars).to(gpu)
    blosum.augment_packed(ref, dof, chain_len=1, augment_frac=0.5, seed=8)
    blosum.augment_packed(ref, dof, chain_len=1, augment_frac=0.5, seed=8)
    want_chars = ref.cpu().numpy()
    assert buf.cpu().numpy().tobytes() == want_chars.tobytes()
    assert out.cpu().numpy().tobytes() == ora.tokenize_packed(want_chars, offs, P, "b", True).tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("chain_len,frac", [(1, 0.5), (3, 1.0)])
def test_cfg5_full_size_one_launch_vs_oracle(gpu, bsq, oracle, chain_len, frac):
    """BASELINE config 5 at FULL size (262 144 x 512 SEB8) through the one-call entry, both layouts: the mutated characters are
    those of bsq_augment_device (the two-call form), and the tokens are the ORACLE's encode of those mutated bytes -- not a
    comparison with the product's own token kernel."""
    import torch
    from bioseq_amd import blosum, capi, synth
    lib = capi.load()
    cfg = synth.CONFIGS["cfg5"]
    n, P = cfg["n"], cfg["padlen"]
    chars, offs = synth.synth_packed(cfg["seed"], n, cfg["lo"], cfg["hi"], cfg["letters"])
    dof = torch.from_numpy(offs).to(gpu)
    tok = bsq.Tokenizer(cfg["key"], cfg["eos"], cfg["bos"], cfg["padchar"])
    ora = oracle.OracleTokenizer(cfg["key"], cfg["eos"], cfg["bos"], cfg["padchar"])
    ref = torch.from_numpy(chars).to(gpu)
    blosum.augment_packed(ref, dof, chain_len=chain_len, augment_frac=frac, seed=1)
    want_chars = ref.cpu().numpy()
    del ref
    ndiff = int((want_chars != chars).sum())
    assert (0.4 * n < ndiff < 0.6 * n) if frac < 1.0 else (n <= ndiff + n // 8 and ndiff <= chain_len * n)
    for bf, knob in ((True, 0), (True, 4), (False, 0)):   # knob 0: this size takes the flag form; 4: the no-wait form + patch launch (round 5)
        want = ora.tokenize_packed(want_chars, offs, P, "b", bf)
        buf = torch.from_numpy(chars).to(gpu)
        capi.check(lib.bsq_tuning_set(b"augment_fused", knob))
        try:
            got = blosum.augment_tokenize_packed(tok, buf, dof, P, "b", bf, chain_len=chain_len, augment_frac=frac, seed=1)
        finally:
            capi.check(lib.bsq_tuning_set(b"augment_fused", 0))
        blosum.check_fused(synchronize=True)
        assert buf.cpu().numpy().tobytes() == want_chars.tobytes(), (chain_len, frac, bf, knob)
        assert got.cpu().numpy().tobytes() == want.tobytes(), (chain_len, frac, bf, knob)
        del buf, got, want


@pytest.mark.gpu
def test_fused_launch_under_contention(gpu, bsq, oracle):
    """The one-launch form rests on the dispatcher starting workgroups in order.  Exercised where that could matter: (a) while a
    second stream keeps the GPU busy with back-to-back multi-GB fills, (b) queued behind 64 small kernels on its own stream,
    (c) both at once, several launches each -- bit-exact with the two-call form + the oracle every time, and no wave gave up."""
    import torch
    from bioseq_amd import blosum, synth
    B, P = 120000, 512
    chars, offs = synth.synth_packed(90210, B, 30, 510, synth.AA)
    dof = torch.from_numpy(offs).to(gpu)
    tok = bsq.Tokenizer("SEB8")
    ora = oracle.OracleTokenizer("SEB8")
    want_c, want_t = {}, {}
    for seed in range(4):
        ref = torch.from_numpy(chars).to(gpu)
        blosum.augment_packed(ref, dof, chain_len=1, augment_frac=0.5, seed=seed)
        want_c[seed] = ref.cpu().numpy()
        want_t[seed] = ora.tokenize_packed(want_c[seed], offs, P, "b", True)
    big = torch.empty(3 << 30, dtype=torch.uint8, device=gpu)
    side = torch.cuda.Stream(device=gpu)
    small = torch.zeros(1024, dtype=torch.float32, device=gpu)
    for mode in ("fills", "queued", "both"):
        for seed in range(4):
            buf = torch.from_numpy(chars).to(gpu)
            torch.cuda.synchronize()
            if mode in ("fills", "both"):
                with torch.cuda.stream(side):
                    for k in range(6):
                        big.fill_(k)
            if mode in ("queued", "both"):
                for _ in range(64):
                    small.add_(1.0)
            got = blosum.augment_tokenize_packed(tok, buf, dof, P, "b", True, chain_len=1, augment_frac=0.5, seed=seed)
            blosum.check_fused(synchronize=True)
            assert buf.cpu().numpy().tobytes() == want_c[seed].tobytes(), (mode, seed)
            assert got.cpu().numpy().tobytes() == want_t[seed].tobytes(), (mode, seed)


@pytest.mark.gpu
def test_fused_wait_expiry_is_loud(gpu, bsq, oracle):
    """Fault injection (knob fused_spins = 1: a token wave polls its flags ONCE): the waves that come too early give up, poison
    their 4 KiB of the output with 0xFF and are counted.  The status call reports BSQ_ERR_FUSED_WAIT, the NEXT one-call entry
    refuses at entry, `check_fused` raises; after `clear_fused_error` everything works again.  Never wrong tokens under BSQ_OK."""
    import torch
    from bioseq_amd import blosum, capi, synth
    lib = capi.load()
    B, P = 200000, 512
    chars, offs = synth.synth_packed(555, B, 30, 510, synth.AA)
    dof = torch.from_numpy(offs).to(gpu)
    tok = bsq.Tokenizer("SEB8")
    ora = oracle.OracleTokenizer("SEB8")
    ref = torch.from_numpy(chars).to(gpu)
    blosum.augment_packed(ref, dof, chain_len=1, augment_frac=0.5, seed=3)
    want = ora.tokenize_packed(ref.cpu().numpy(), offs, P, "b", True)
    capi.check(lib.bsq_tuning_set(b"fused_spins", 1))
    capi.check(lib.bsq_tuning_set(b"augment_fused", 2))  # the flag form (since round 5 the default form has no wait that could expire)
    try:
        buf = torch.from_numpy(chars).to(gpu)
        got = blosum.augment_tokenize_packed(tok, buf, dof, P, "b", True, chain_len=1, augment_frac=0.5, seed=3)
        torch.cuda.synchronize()
        nfail = ctypes.c_uint32(0)
        st = lib.bsq_fused_status(ctypes.byref(nfail))
        g = got.cpu().numpy()
        if nfail.value == 0:  # every wave found its flags up at its single poll (possible in principle): then the result must be right
            assert st == 0 and g.tobytes() == want.tobytes()
            pytest.skip("no token wave came early enough to give up on this box")
        assert st == 8  # BSQ_ERR_FUSED_WAIT
        flat = g.reshape(-1, 4096).view(np.uint8)
        poisoned = (flat == 0xFF).all(axis=1)
        assert int(poisoned.sum()) == nfail.value                      # one 4-KiB chunk per wave that gave up, all of it 0xFF
        assert (flat[~poisoned] == want.reshape(-1, 4096).view(np.uint8)[~poisoned]).all()  # every other chunk is right
        with pytest.raises(RuntimeError, match="gave up waiting"):
            blosum.check_fused()
        with pytest.raises(RuntimeError, match="gave up waiting"):     # sticky: the next call refuses at entry
            blosum.augment_tokenize_packed(tok, torch.from_numpy(chars).to(gpu), dof, P, "b", True, chain_len=1, augment_frac=0.5, seed=3)
    finally:
        capi.check(lib.bsq_tuning_set(b"fused_spins", 0))
        capi.check(lib.bsq_tuning_set(b"augment_fused", 0))
        blosum.clear_fused_error()
    buf = torch.from_numpy(chars).to(gpu)
    got = blosum.augment_tokenize_packed(tok, buf, dof, P, "b", True, chain_len=1, augment_frac=0.5, seed=3)
    blosum.check_fused(synchronize=True)
    assert got.cpu().numpy().tobytes() == want.tobytes()


def test_integer_acceptance_thresholds_equal_the_twins_float_test():
    """The kernel accepts a position iff lo32 <= threshold[row]; the twin (and the law) iff lo32 * 2**-32 < 1 - p_self.  Same truth
    value for every 32-bit word: checked at and around every row's boundary and at the extremes (CPU, no GPU needed)."""
    from bioseq_amd import blosum, capi
    lib = capi.load()
    thr = (ctypes.c_uint32 * 21)()
    capi.check(lib.bsq_blosum62_accept_thresholds(thr))
    for row in range(21):
        pself = float(blosum.normrows[row, row]) if row < 20 else 0.0
        t = int(thr[row])
        probes = {0, 1, 2 ** 31, 2 ** 32 - 2, 2 ** 32 - 1} | {min(max(t + d, 0), 2 ** 32 - 1) for d in range(-3, 4)}
        for lo in probes:
            assert (lo <= t) == (lo * 2.0 ** -32 < 1.0 - pself), (row, lo, t)
