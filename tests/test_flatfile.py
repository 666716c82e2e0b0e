"""FlatFile: format parity with the reference's writer, the reference's access surface,
and the zero-copy packed-batch path into the encoder."""
import os

import numpy as np
import pytest

EXPECT = [b"MKVLAAGIVGLLLAQPSNA", b"", b"ACDEFGHIKLMNPQRSTVWYacdef", b"ACGTN", b"WWWW"]


def test_reads_the_reference_written_file(golden_dir):
    from bioseq_amd.flatfile import FlatFile
    ff = FlatFile(os.path.join(golden_dir, "flatfile_small.ff"))
    assert len(ff) == ff.nseqs() == ff.size() == 5 and ff.maxseqlen == ff.max_seq_len == 25
    assert ff.seq_offset() == (5 + 2) * 8
    assert ff.indptr().dtype == np.uint64 and ff.indptr().tolist() == [0, 19, 19, 44, 49, 53]
    assert [bytes(x.seq) for x in ff] == EXPECT and [bytes(x.sequence) for x in ff] == EXPECT
    assert isinstance(ff[0], bytearray) and bytes(ff[-1]) == b"WWWW" and bytes(ff.access(3)) == b"ACGTN"
    assert [bytes(x) for x in ff.access(0, 3)] == EXPECT[:3] and [bytes(x) for x in ff[1:4]] == EXPECT[1:4]
    assert [bytes(x) for x in ff.access(0, 5, 2)] == EXPECT[::2]
    assert [bytes(x) for x in ff[np.array([4, 0])]] == [EXPECT[4], EXPECT[0]]
    with pytest.raises(IndexError):
        ff.access(5)
    chars, offs = ff.packed(2, 5)
    assert offs.tolist() == [0, 25, 30, 34] and chars.tobytes() == b"".join(EXPECT[2:5])


def test_writer_is_byte_identical_to_the_reference(golden_dir, tmp_path):
    from bioseq_amd.flatfile import FlatFile
    out = str(tmp_path / "mine.ff")
    ff = FlatFile(os.path.join(golden_dir, "flatfile_small.fa"), out)
    assert ff.path == out
    assert open(out, "rb").read() == open(os.path.join(golden_dir, "flatfile_small.ff"), "rb").read()
    # gzip input and the default output name
    import gzip
    import shutil
    gz = str(tmp_path / "x.fa.gz")
    with open(os.path.join(golden_dir, "flatfile_small.fa"), "rb") as a, gzip.open(gz, "wb") as b:
        shutil.copyfileobj(a, b)
    ff2 = FlatFile(gz, "")
    assert ff2.path == gz + ".ff" and [bytes(x.seq) for x in ff2] == EXPECT


@pytest.mark.gpu
def test_flatfile_to_encoder_zero_copy(gpu, bsq, oracle, golden_dir, tmp_path):
    from bioseq_amd import synth
    from bioseq_amd.flatfile import FlatFile, write_flatfile
    chars, offs = synth.synth_packed(77, 3000, 0, 300, synth.AA)
    path = write_flatfile(synth.unpack(chars, offs), str(tmp_path / "big.ff"))
    ff = FlatFile(path)
    tok, ora = bsq.Tokenizer("AMINO20", 1, 1, 0), oracle.OracleTokenizer("AMINO20", 1, 1, 0)
    P = ff.maxseqlen + 2
    for a, b in ((0, 3000), (100, 1124), (2990, 3000)):
        c, o = synth.unpack(chars, offs)[a:b], None
        exp_t = ora.batch_tokenize(c, padlen=P, batch_first=True)
        exp_o = ora.batch_onehot_encode(c, padlen=P, destchar="f")
        assert ff.batch_tokenize(tok, a, b).tobytes() == exp_t.tobytes()                         # mapped file -> staged
        assert ff.batch_tokenize(tok, a, b, device=gpu).cpu().numpy().tobytes() == exp_t.tobytes()  # resident store
        assert ff.batch_onehot_encode(tok, a, b, destchar="f", device=gpu).cpu().numpy().tobytes() == exp_o.tobytes()
        # same result as the reference's route: access() -> list of bytearrays -> batch_tokenize
        assert tok.batch_tokenize(ff.access(a, b), padlen=P, batch_first=True).tobytes() == exp_t.tobytes()


# ---------------------------------------------------------------- native FASTX reader vs the compiled reference (CPU)

def _random_fastx(rng):
    """Adversarial FASTA / FASTQ text: CRLF lines, blank lines, '>' / '@' / '+' inside lines, multi-line sequences,
    quality strings that are too short / too long / missing, junk before the first header, no final newline."""
    alpha = b"ACGTNacgtn*MKVLW -\t"
    out = bytearray()
    if rng.random() < 0.3:
        out += rng.choice([b"junk line\n", b"\n\n", b"xx>inline header\n", b"# comment @ here\n"])
    for _ in range(int(rng.integers(0, 12))):
        nl = b"\r\n" if rng.random() < 0.25 else b"\n"
        kind = rng.random()
        name = bytes(rng.choice(list(b"abcXYZ09_|"), size=int(rng.integers(0, 6))).astype(np.uint8))
        comment = b"" if rng.random() < 0.5 else b" some comment > with @ signs"
        lines = []
        for _ in range(int(rng.integers(0, 4))):
            body = bytes(rng.choice(list(alpha), size=int(rng.integers(0, 30))).astype(np.uint8))
            if rng.random() < 0.1:
                body = b"" if rng.random() < 0.5 else b"\r"
            if body[:1] in (b">", b"@", b"+"):
                body = b"A" + body
            lines.append(body)
        if kind < 0.55:                                          # FASTA
            out += b">" + name + comment + nl + b"".join(l + nl for l in lines)
        else:                                                    # FASTQ
            seqlen = sum(len(l) for l in lines)
            q = rng.random()
            qlen = seqlen if q < 0.7 else max(0, seqlen + int(rng.integers(-3, 4)))
            qual = bytes(rng.choice(list(b"IIII#!>@+5"), size=qlen).astype(np.uint8))
            if qual[:1] in (b">", b"@") and rng.random() < 0.5:
                qual = b"I" + qual[1:]
            out += b"@" + name + comment + nl + b"".join(l + nl for l in lines) + b"+" + (name if rng.random() < 0.5 else b"") + nl
            if rng.random() < 0.9:
                cut = int(rng.integers(0, len(qual) + 1)) if rng.random() < 0.3 else len(qual)
                out += qual[:cut] + nl + (qual[cut:] + nl if cut < len(qual) else b"")
    if out and rng.random() < 0.3:
        out = out.rstrip(b"\r\n")
    return bytes(out)


def test_native_fastx_reader_equals_the_compiled_reference(oracle, tmp_path):
    """Differential: bioseq_amd.FlatFile(fastx, out) / getstats against oracle/_ref = the reference's own fxstats.cpp
    (kseq + zlib) on 400 random texts, plain and gzipped.  Skipped where the reference build is absent."""
    import gzip
    import bioseq_amd
    from bioseq_amd.flatfile import FlatFile
    ref = oracle.load_reference()
    if ref is None or not hasattr(ref, "FlatFile"):
        pytest.skip("oracle/_ref (the compiled reference) is not present")
    rng = np.random.default_rng(2026)
    nonempty = 0
    for case in range(400):
        text = _random_fastx(rng)
        src = str(tmp_path / ("c%d.fx" % case)) + (".gz" if case % 5 == 0 else "")
        with (gzip.open if src.endswith(".gz") else open)(src, "wb") as f:
            f.write(text)
        want_lens = ref.getstats([src])[0]
        got_lens = bioseq_amd.getstats([src])[0]
        assert got_lens.dtype == np.uint64 and got_lens.tolist() == want_lens.tolist(), (case, text)
        a, b = str(tmp_path / "ref.ff"), str(tmp_path / "mine.ff")
        ref.FlatFile(src, a)
        mine = FlatFile(src, b)
        assert open(b, "rb").read() == open(a, "rb").read(), (case, text)
        assert len(mine) == len(want_lens)
        nonempty += len(want_lens) > 0
        os.remove(src)
    assert nonempty > 200
    with pytest.raises(RuntimeError, match="failed to open"):
        FlatFile(str(tmp_path / "does_not_exist.fa"), str(tmp_path / "o.ff"))


def test_native_fastx_reader_long_lines_and_truncated_gzip(oracle, tmp_path):
    """Differential against the compiled reference on what a streaming parser of untrusted text gets wrong first: lines around the
    reader's buffer sizes (16 383 / 16 384 / 16 385 / 131 072 bytes, FASTA and FASTQ, with and without a final newline) and gzip
    streams cut off at every kind of place.  Runs under ASan / UBSan in scripts/asan_host.sh."""
    import gzip
    import bioseq_amd
    from bioseq_amd.flatfile import FlatFile
    ref = oracle.load_reference()
    if ref is None or not hasattr(ref, "FlatFile"):
        pytest.skip("oracle/_ref (the compiled reference) is not present")
    rng = np.random.default_rng(16384)

    def same(src, tag):
        want = ref.getstats([src])[0]
        got = bioseq_amd.getstats([src])[0]
        assert got.tolist() == want.tolist(), tag
        a, b = str(tmp_path / "ref.ff"), str(tmp_path / "mine.ff")
        ref.FlatFile(src, a)
        FlatFile(src, b)
        assert open(b, "rb").read() == open(a, "rb").read(), tag

    texts = []
    for n in (16383, 16384, 16385, 32767, 32768, 131072):
        body = bytes(rng.choice(list(b"ACGTN"), size=n).astype(np.uint8))
        for end in (b"\n", b""):
            texts.append(b">long " + b"c" * (n % 7) + b"\n" + body + end)                                     # one line of n bytes
            texts.append(b">a\nAC\n>two\n" + body[: n // 2] + b"\n" + body[n // 2:] + end)                    # split over two lines
            texts.append(b"@q\n" + body + b"\n+\n" + b"I" * n + end)                                          # FASTQ, quality as long
            texts.append(b">" + b"h" * n + b"\n" + body[:100] + b"\n>next\nACGT" + end)                        # a HEADER of n bytes
    for i, text in enumerate(texts):
        src = str(tmp_path / ("l%d.fx" % i))
        with open(src, "wb") as f:
            f.write(text)
        same(src, ("plain", i))
        if i % 3 == 0:
            with gzip.open(src + ".gz", "wb", compresslevel=1) as f:
                f.write(text)
            same(src + ".gz", ("gz", i))
            os.remove(src + ".gz")
        os.remove(src)
    # truncated gzip members: both readers see the same prefix of records (or both fail)
    text = b"".join(b">r%d\n" % k + bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(1, 300))).astype(np.uint8)) + b"\n" for k in range(400))
    whole = gzip.compress(text, compresslevel=6)
    for j, cut in enumerate([1, 9, 10, 11, 18, 64, len(whole) // 3, len(whole) // 2, len(whole) - 9, len(whole) - 8, len(whole) - 1]):
        src = str(tmp_path / ("t%d.fa.gz" % j))
        with open(src, "wb") as f:
            f.write(whole[:cut])
        try:
            want = ref.getstats([src])[0].tolist()
        except RuntimeError:
            with pytest.raises(RuntimeError):
                bioseq_amd.getstats([src])
            continue
        assert bioseq_amd.getstats([src])[0].tolist() == want, ("truncated", cut)
        os.remove(src)


def test_native_fastx_reader_streams_a_large_fastq(tmp_path):
    """200 000 reads of 150 bases (FASTQ-shaped, BASELINE config 4's source format, gzipped): the writer keeps only the offsets
    in memory; the result round-trips through the FlatFile surface."""
    import gzip
    import resource
    from bioseq_amd.flatfile import FlatFile, fastx_to_flatfile
    rng = np.random.default_rng(4)
    n, L = 200000, 150
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, L))]
    src = str(tmp_path / "reads.fq.gz")
    with gzip.open(src, "wb", compresslevel=1) as f:
        qual = b"I" * L
        f.write(b"".join(b"@r%d\n" % i + bases[i].tobytes() + b"\n+\n" + qual + b"\n" for i in range(n)))
    before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    out, nseqs, longest = fastx_to_flatfile(src, str(tmp_path / "reads.ff"))
    grown_mb = (resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - before) / 1024.0
    assert (nseqs, longest) == (n, L)
    assert grown_mb < 64, grown_mb                       # the 30 MB of sequence bytes are never held in memory
    ff = FlatFile(out)
    assert len(ff) == n and ff.maxseqlen == L
    c, o = ff.packed()
    assert c.tobytes() == bases.tobytes() and (np.diff(o) == L).all()
    assert not os.path.exists(out + ".seq.tmp")
