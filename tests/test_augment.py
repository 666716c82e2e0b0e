"""BLOSUM62 augmentation.  CPU: the transition table is bit-identical to the
reference's.  GPU: the kernel equals a numpy twin of its counter-based algorithm bit for bit, keeps the
reference's invariants (<= chain_len positions change, new != old, unknown residues use the X row,
augment_frac) and reproduces normrows statistically (chi-square)."""
import hashlib
import os

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
    for chain_len, frac, seed in ((1, 1.0, 7), (3, 0.5, 123456789012345), (2, 1.5, 2 ** 63 + 5)):
        d = blosum.augment_packed(torch.from_numpy(chars).to(gpu), torch.from_numpy(offs).to(gpu), chain_len, frac, seed)
        got = d.cpu().numpy()
        exp = twin(chars, offs, chain_len, frac, seed, blosum.normrows)
        assert got.tobytes() == exp.tobytes()
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
